#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c25; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
( cd /tmp && export TMPDIR=/tmp && CNMFE_BENCH_FORCE_COLLECTIVES=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --config c4 --steps 8 --warmup 3 > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/trace.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r4c25/summary.txt
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
t0 = rows[0][0]
print("kernels", len(rows), "span %.1f ms" % ((rows[-1][1] - t0) / 1e6))
print("--- kernels longer than 3 ms")
for s, e, n in rows:
    if e - s > 3e6: print("%9.1f ms  %7.2f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n[:80]))
print("--- gaps longer than 5 ms")
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if s1 - e0 > 5e6: print("%9.1f ms  gap %7.2f ms  after %s | before %s" % ((e0 - t0) / 1e6, (s1 - e0) / 1e6, n0[:50], n1[:50]))
PY
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4c25/bench.json").read().strip().splitlines()[-1]); print("forced under rocprof ms/step %.2f" % j["ms_per_step"])
PY
rm -rf $O/trace
