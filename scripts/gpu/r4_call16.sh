#!/bin/bash
# native search masks + packed solve: benches, host timeline, kernel timeline; parity suites
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c16; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
X="--no-extras --no-cpu-baseline"
timeout 300 python bench.py $X > $O/c3.json 2> $O/c3.err
timeout 300 python bench.py $X > $O/c3_b.json 2> $O/c3_b.err
timeout 300 python bench.py $X --config c4 --steps 5 > $O/c4.json 2> $O/c4.err
timeout 300 python bench.py $X --config c2 > $O/c2.json 2> $O/c2.err
timeout 300 python bench.py $X --deconv > $O/deconv.json 2> $O/deconv.err
timeout 300 python bench.py $X --bg-ssub 2 --deconv --alg hals_thresh > $O/demo.json 2> $O/demo.err
python scripts/host_timeline.py > $O/host_timeline.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/gap_analysis.py "$f" seq > $O/gap_analysis_c3.txt 2>&1
rm -rf $O/trace
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_virtual.py tests/test_gpu_kchange.py tests/test_gpu_packed.py -x -q > $O/test_a.txt 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/test_a.txt
python - <<'PY' | tee -a gpurun_out/r4c16/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c16/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j.get("kernels_ms_per_step", {})
        print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], "sum", j.get("kernel_sum_ms_per_step"), "solve", k.get("bg_ring_solve"))
    except Exception as e:
        print(f, "ERR", e)
PY
head -16 $O/gap_analysis_c3.txt
tail -14 $O/host_timeline.txt
