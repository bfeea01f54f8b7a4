#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c18; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
timeout 300 python -m pytest tests/test_gpu_packed.py -x -q > $O/test_packed.txt 2>&1; echo "packed tests rc=$?"
tail -3 $O/test_packed.txt
timeout 300 python scripts/host_timeline.py --patch 128 --as-rank-of 8 --iters 4 --cprofile > $O/host_timeline_rank_of_8.txt 2>&1
timeout 300 python scripts/host_timeline.py --patch 128 --as-rank-of 2 --iters 4 > $O/host_timeline_rank_of_2.txt 2>&1
timeout 300 python bench.py --no-extras --no-cpu-baseline > $O/c3.json 2> $O/c3.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4c18/c3.json").read().strip().splitlines()[-1]); print("c3 ms/step %.2f" % j["ms_per_step"], j["kernels_ms_per_step"].get("bg_ring_solve"))
PY
head -60 $O/host_timeline_rank_of_8.txt
