#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c40; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_packed.py tests/test_gpu_virtual.py -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt
X="--no-extras --no-cpu-baseline"
for i in 1 2 3; do timeout 300 python bench.py $X > $O/c3_$i.json 2> $O/c3_$i.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c40/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], "kernel sum", j["kernel_sum_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
