"""Idle gaps of the GPU between consecutive kernels of one iteration, from a rocprofv3 --kernel-trace CSV: python scripts/gap_analysis.py <kernel_trace.csv> [n_last_kernels]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
# one iteration = from one k_ring_pmax (the kernel a background fit ends with) to the next
idx = [i for i, r in enumerate(rows) if "k_ring_pmax" in r[2]]
a, b = idx[-2], idx[-1]
it = rows[a:b]
span = (it[-1][1] - it[0][0]) / 1e6
busy = sum(e - s for s, e, _ in it) / 1e6
print("iteration span %.2f ms (to the end of its last kernel; the next build_bf starts %.2f ms later), kernels busy %.2f ms, %d kernels" % (span, (rows[b][0] - it[-1][1]) / 1e6, busy, len(it)))
gaps = []
for i in range(1, len(it) + 1):
    nxt = it[i] if i < len(it) else rows[b]
    g = (nxt[0] - it[i - 1][1]) / 1e6
    gaps.append((g, it[i - 1][2][:50], nxt[2][:50]))
for g, p, n in sorted(gaps, reverse=True)[:14]:
    print("%7.3f ms  after %-50s before %s" % (g, p, n))
print("sum of gaps %.2f ms; gaps < 20 us: %d totalling %.2f ms" % (sum(g for g, _, _ in gaps), sum(1 for g, _, _ in gaps if g < 0.02), sum(g for g, _, _ in gaps if g < 0.02)))
if len(sys.argv) > 2 and sys.argv[2] == "seq":
    t0 = it[0][0]
    prev = None
    for s_, e_, n_ in it:
        gap = (s_ - prev) / 1e3 if prev else 0.0
        print("%8.3f ms  +%7.1f us gap  %8.1f us  %s" % ((s_ - t0) / 1e6, gap, (e_ - s_) / 1e3, n_[:70]))
        prev = e_
