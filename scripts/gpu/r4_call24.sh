#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c24; mkdir -p $O
export PYTHONUNBUFFERED=1
CNMFE_TRACE_GATHER=1 timeout 300 python scripts/host_timeline.py --patch 128 --iters 8 --force-collectives > $O/ht_forced.txt 2>&1
grep -E "^iteration|gather\]" $O/ht_forced.txt | head -150
