#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c19; mkdir -p $O
export PYTHONUNBUFFERED=1
CNMFE_OPTS=host_trace=1 timeout 300 python scripts/host_timeline.py --patch 128 --as-rank-of 8 --iters 3 > $O/ht8.txt 2> $O/ht8.err
grep -n "host_trace" $O/ht8.err | tail -90
tail -22 $O/ht8.txt
