#!/bin/bash
# (a) how fast the lanes' scratch settles (fenced warm-up steps of c4 on three lanes), (b) a kernel trace of c4 on three lanes: how much of an iteration has kernels of two streams in flight
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_lanes.py -x -q 2>&1 | grep -a "passed\|failed" | tail -2
CNMFE_BENCH_LANES=3 timeout 280 python bench.py --config c4 --steps 6 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c4 lanes', d['config'].get('lanes_per_rank'), round(d['ms_per_step'],2), d['first_iteration']['warmup_steps_ms'])"
cd /tmp && export TMPDIR=/tmp
for l in 1 3; do
rm -rf /tmp/tl$l && CNMFE_BENCH_LANES=$l timeout 280 rocprofv3 --kernel-trace -f csv -d /tmp/tl$l -o tl -- python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 4 --warmup 6 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find /tmp/tl$l -name "*kernel_trace.csv" | head -1)
python - "$f" $l <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id") or r.get("Stream_Id") or "") for r in rows if "cnmfe::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]))
idx = [i for i, e in enumerate(ev) if "k_stitch_finish" in e[2]]
a, b = idx[-3] + 1, idx[-1] + 1                      # the last two iterations
pts = []
for s, e, n, q in ev[a:b]:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
t0, t1 = ev[a][0], max(e for _, e, _, _ in ev[a:b])
cur = 0; last = t0; busy = [0, 0, 0, 0, 0]
for t, dlt in pts:
    busy[min(cur, 4)] += t - last; last = t; cur += dlt
span = t1 - t0
queues = sorted(set(q for _, _, _, q in ev[a:b]))
print("lanes %s: two iterations span %.2f ms, %d launches on %d queue(s); in flight 0 / 1 / 2 / 3 / 4+ kernels: %s %% of the span; sum of kernel durations %.2f ms" % (
    sys.argv[2], span / 1e6, b - a, len(queues), " / ".join("%.1f" % (100.0 * x / span) for x in busy), sum(e - s for s, e, _, _ in ev[a:b]) / 1e6))
PY
done
