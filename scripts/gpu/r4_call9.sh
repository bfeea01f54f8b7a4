#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c9; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_virtual.py tests/test_gpu_parity.py tests/test_gpu_kchange.py -x -q > $O/test_some.txt 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -5 $O/test_some.txt >> $O/summary.txt
for rep in 1 2; do for v in defer1 defer0; do
  CNMFE_OPTS=solve_defer=${v#defer} CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  CNMFE_OPTS=solve_defer=${v#defer} CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 --config c2 > $O/bench_c2_${v}_$rep.json 2> /dev/null
done; done
python - <<'PY' >> gpurun_out/r4c9/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c9/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j["kernels_ms_per_step"]
        print(f.split("bench_")[1][:-5], "ms/step %.2f" % j["ms_per_step"], "sum", j["kernel_sum_ms_per_step"], "solve", k.get("bg_ring_solve"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
