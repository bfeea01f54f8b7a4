#!/bin/bash
# round 4, GPU call 2: the sweep-free residual (vproj.hip): its own tests first, then the whole GPU suite, then the headline bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c2; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_virtual.py -x -q > $O/test_virtual.txt 2>&1; echo "virtual tests rc=$?" | tee -a $O/summary.txt; tail -30 $O/test_virtual.txt >> $O/summary.txt
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?" >> $O/summary.txt
CNMFE_OPTS=r1_virtual=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_c3_sweep.json 2> $O/bench_c3_sweep.err; echo "bench c3 (sweep) rc=$?" >> $O/summary.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/test_all.txt 2>&1; echo "all tests rc=$?" | tee -a $O/summary.txt; tail -30 $O/test_all.txt >> $O/summary.txt
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --config c4 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?" >> $O/summary.txt
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --config c2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?" >> $O/summary.txt
python - <<'PY' >> gpurun_out/r4c2/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c2/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j.get("kernels_ms_per_step", {})
        print(f.split("bench_")[1][:-5], "ms/step %.2f" % j["ms_per_step"], "sum", j.get("kernel_sum_ms_per_step"), {a: round(b, 3) for a, b in sorted(k.items(), key=lambda x: -x[1])[:16]})
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
