#!/bin/bash
# packed ring solve: its own tests, the ring-fit edge tests and the parity suite under the new default, then A/B benches (solve_packed 1 / 0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c14; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
timeout 600 python -m pytest tests/test_gpu_packed.py -x -q > $O/test_packed.txt 2>&1; echo "packed tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/test_packed.txt
X="--no-extras --no-cpu-baseline"
for v in 1 0; do
  CNMFE_OPTS=solve_packed=$v timeout 300 python bench.py $X > $O/c3_packed$v.json 2> $O/c3_packed$v.err; echo "bench c3 packed=$v rc=$?" | tee -a $O/summary.txt
done
CNMFE_OPTS=solve_packed=1 timeout 300 python bench.py $X --config c4 --steps 5 > $O/c4_packed1.json 2> $O/c4_packed1.err
CNMFE_OPTS=solve_packed=1 timeout 300 python bench.py $X --config c2 > $O/c2_packed1.json 2> $O/c2_packed1.err
CNMFE_OPTS=solve_packed=1 timeout 300 python bench.py $X --bg-ssub 2 > $O/ssub2_packed1.json 2> $O/ssub2_packed1.err
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_virtual.py -x -q > $O/test_rest.txt 2>&1; echo "edge/parity/virtual tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/test_rest.txt
python - <<'PY' | tee -a gpurun_out/r4c14/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c14/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j.get("kernels_ms_per_step", {})
        print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], "sum", j.get("kernel_sum_ms_per_step"), {n: round(v, 3) for n, v in list(k.items())[:8]})
    except Exception as e:
        print(f, "ERR", e)
PY
