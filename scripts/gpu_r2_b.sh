#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python scripts/solve_ab.py --cfg small --radius 18 --modes 2,5 2>&1 | tail -n 3
timeout 600 python scripts/solve_ab.py --cfg c3 --modes 2,5 --probes 0,1,2,4,7 2>&1 | grep -v amdgpu.ids
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_b.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print(d["kernels_ms_per_step"])
PY
