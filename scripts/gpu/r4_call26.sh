#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c26; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0 CNMFE_BENCH_FORCE_COLLECTIVES=1
X="--no-extras --no-cpu-baseline --config c4 --steps 10 --warmup 4"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $X > $O/$name.json 2> $O/$name.err; python - "$name" <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/r4c26/%s.json" % sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], "ms/step %.2f" % j["ms_per_step"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
run base A=1
run async0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0
run monitor0 TORCH_NCCL_ENABLE_MONITORING=0
run blocking TORCH_NCCL_BLOCKING_WAIT=1
run both0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0
run nohb TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_DUMP_ON_TIMEOUT=0 TORCH_NCCL_DESYNC_DEBUG=0 TORCH_NCCL_ENABLE_TIMING=0
