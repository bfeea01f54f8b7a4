#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c32; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { name=$1; shift; env "$@" timeout 300 python scripts/host_timeline.py --patch 128 --iters 10 --force-collectives > $O/$name.txt 2> $O/$name.err; echo "$name:"; grep -E "^iteration" $O/$name.txt | awk '{printf "%s ", $3}'; echo; }
run nopin GPU_PINNED_MIN_XFER_SIZE=1073741824 GPU_PINNED_XFER_SIZE=64
run malloc_arena MALLOC_ARENA_MAX=1
run omp1 OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1
