#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c38; mkdir -p $O
export PYTHONUNBUFFERED=1
CNMFE_OPTS=host_trace=1 timeout 300 python scripts/host_timeline.py --iters 4 > $O/ht.txt 2> $O/ht.err
tail -14 $O/ht.txt
grep "host_trace" $O/ht.err | tail -42
