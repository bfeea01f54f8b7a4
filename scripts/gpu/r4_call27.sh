#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c27; mkdir -p $O
export PYTHONUNBUFFERED=1
CNMFE_OPTS=host_trace=2 timeout 300 python scripts/host_timeline.py --patch 128 --iters 8 --force-collectives > $O/ht_forced.txt 2> $O/ht_forced.err
grep -E "^iteration" $O/ht_forced.txt
grep -E "launch .*flush" $O/ht_forced.err | head -40
grep -c "host_trace" $O/ht_forced.err
