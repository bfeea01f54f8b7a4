#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c31; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { name=$1; shift; env "$@" timeout 300 python scripts/host_timeline.py --patch 128 --iters 10 --force-collectives > $O/$name.txt 2> $O/$name.err; echo "$name:"; grep -E "^iteration" $O/$name.txt | awk '{printf "%s ", $3}'; echo; }
run fr0 TORCH_FR_BUFFER_SIZE=0 TORCH_NCCL_TRACE_BUFFER_SIZE=0
run fr0_nocpp TORCH_FR_BUFFER_SIZE=0 TORCH_NCCL_TRACE_BUFFER_SIZE=0 TORCH_NCCL_TRACE_CPP_STACK=0 TORCH_NCCL_ENABLE_TIMING=0
run switch1 PYTHON_SWITCH=1
python - <<'PY'
import torch, os
print(torch.__version__)
for k, v in sorted(os.environ.items()):
    if "NCCL" in k or "TORCH" in k or "RCCL" in k or "HSA" in k: print(k, v)
PY
