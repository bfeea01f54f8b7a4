#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c36; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
timeout 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_kchange.py -x -q -k "sn or deconv or noise or long" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt
X="--no-extras --no-cpu-baseline"
timeout 300 python bench.py $X --demo-sequence > $O/demo_seq.json 2> $O/demo_seq.err
timeout 300 python bench.py $X --deconv > $O/deconv.json 2> $O/deconv.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4c36/demo_seq.json").read().strip().splitlines()[-1]); print("demo sequence %.3f s" % j["value"], {k: v for k, v in list(j["kernels_ms_total"].items())[:4]})
j = json.loads(open("gpurun_out/r4c36/deconv.json").read().strip().splitlines()[-1]); print("deconv ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in list(j["kernels_ms_per_step"].items())[:3]})
PY
