#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c28; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python scripts/host_timeline.py --patch 128 --iters 8 --force-collectives > $O/a.txt 2> $O/a.err
grep -E "^iteration|affinity|threads" $O/a.txt
RESET_AFFINITY=1 timeout 300 python scripts/host_timeline.py --patch 128 --iters 8 --force-collectives > $O/b.txt 2> $O/b.err
grep -E "^iteration|affinity|threads" $O/b.txt
NCCL_IGNORE_CPU_AFFINITY=1 timeout 300 python scripts/host_timeline.py --patch 128 --iters 8 --force-collectives > $O/c.txt 2> $O/c.err
grep -E "^iteration|affinity|threads" $O/c.txt
nproc; cat /proc/loadavg
