"""Idle gaps of the GPU between consecutive kernels of one iteration, from a rocprofv3 --kernel-trace CSV:
   python scripts/gap_analysis.py <kernel_trace.csv> [seq|agg] [background fits per iteration (patches), default 1]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
# one iteration = from one k_ring_pmax (the kernel a background fit ends with) to the next
idx = [i for i, r in enumerate(rows) if "k_ring_pmax" in r[2]]
npi = int(sys.argv[3]) if len(sys.argv) > 3 else 1
a, b = idx[-1 - npi], idx[-1]
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
if len(sys.argv) > 2 and sys.argv[2] == "agg":
    import collections
    short = lambda n: n.split("(")[0].replace("void ", "").replace("cnmfe::", "")[:40]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for g, p, n in gaps:
        agg[(short(p), short(n))][0] += g; agg[(short(p), short(n))][1] += 1
    print("gaps by (kernel before, kernel after), summed over the iteration:")
    for (p, n), (g, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
        print("%8.3f ms in %4d gaps (%.1f us each)  after %-40s before %s" % (g, c, 1e3 * g / c, p, n))
    kk = collections.defaultdict(lambda: [0.0, 0])
    for s_, e_, n_ in it:
        kk[short(n_)][0] += (e_ - s_) / 1e6; kk[short(n_)][1] += 1
    print("kernels:")
    for n, (t, c) in sorted(kk.items(), key=lambda x: -x[1][0])[:40]:
        print("%8.3f ms in %4d launches (%.1f us each)  %s" % (t, c, 1e3 * t / c, n))
