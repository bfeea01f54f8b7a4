#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c29; mkdir -p $O
export PYTHONUNBUFFERED=1
TRACE_SLOW_PY=1 timeout 300 python scripts/host_timeline.py --patch 128 --iters 10 --force-collectives > $O/a.txt 2> $O/a.err
grep -E "^iteration|slow python" $O/a.txt | head -60
