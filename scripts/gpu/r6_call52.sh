#!/bin/bash
# kernel timeline of one rank's iterations: where the device idles
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rk && timeout 280 rocprofv3 --kernel-trace -f csv -d /tmp/rk -o rk -- python $GRAFT_REPO_ROOT/scripts/rank_load.py --world 8 --steps 6 > $GRAFT_REPO_ROOT/gpurun_out/rank_tl.log 2>&1
f=$(find /tmp/rk -name "*kernel_trace.csv" | head -1)
python - "$f" > $GRAFT_REPO_ROOT/gpurun_out/rank_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows))
# last two iterations: find the k_ymean / first kernel of fit ... simply print the last 400 kernels with gaps
ev = ev[-420:]
t0 = ev[0][0]; busy = 0; prev_end = ev[0][0]
for s, e, n in ev:
    gap = (s - prev_end) / 1e3
    print("%9.1f us  +%7.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n))
    prev_end = max(prev_end, e)
PY
wc -l $GRAFT_REPO_ROOT/gpurun_out/rank_timeline.txt
