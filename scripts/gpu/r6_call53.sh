#!/bin/bash
# kernel timeline of the headline iteration (c3, one patch): every launch and every gap of the last timed iteration
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && timeout 280 rocprofv3 --kernel-trace -f csv -d /tmp/tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/c3_tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" > $GRAFT_REPO_ROOT/gpurun_out/c3_timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
idx = [i for i, e in enumerate(ev) if "k_stitch_finish" in e[2]]
print("stitch_finish at", idx[-8:])
for k in (-3, -2):
    a, b = idx[k - 1] + 1, idx[k] + 1
    t0 = ev[a][0]; prev = ev[a - 1][1]; busy = 0
    for s, e, n in ev[a:b]:
        n = re.sub(r"\(.*", "", n).replace("cnmfe::", "").replace("void ", "")
        g = (s - prev) / 1e3
        print("%7.0f %s%7.0f %s" % ((s - t0) / 1e3, ("gap%5.0f " % g) if g > 15 else "         ", (e - s) / 1e3, n[:50]))
        prev = max(prev, e); busy += e - s
    print("iteration: span %.0f us, busy %.0f us, %d launches" % ((ev[b - 1][1] - ev[a - 1][1]) / 1e3, busy / 1e3, b - a))
PY
tail -1 $GRAFT_REPO_ROOT/gpurun_out/c3_timeline.txt
