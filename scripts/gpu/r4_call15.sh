#!/bin/bash
# packed solve: where the iteration's time goes now (kernel timeline, host timeline), the solve's phases by probe
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c15; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
timeout 300 python -m pytest tests/test_gpu_packed.py -x -q > $O/test_packed.txt 2>&1; echo "packed tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/test_packed.txt
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/gap_analysis.py "$f" seq > $O/gap_analysis_c3_packed.txt 2>&1
rm -rf $O/trace
python scripts/host_timeline.py > $O/host_timeline.txt 2>&1
X="--no-extras --no-cpu-baseline"
for pr in 0 8 2 10; do
  CNMFE_OPTS=solve_probe=$pr timeout 300 python bench.py $X --steps 5 > $O/c3_probe$pr.json 2> $O/c3_probe$pr.err
done
python - <<'PY' | tee -a gpurun_out/r4c15/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c15/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j.get("kernels_ms_per_step", {})
        print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], "sum", j.get("kernel_sum_ms_per_step"), "solve", k.get("bg_ring_solve"))
    except Exception as e:
        print(f, "ERR", e)
PY
head -24 $O/gap_analysis_c3_packed.txt
tail -40 $O/host_timeline.txt
