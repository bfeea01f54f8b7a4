#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c30; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python scripts/host_timeline.py --patch 128 --iters 10 --force-collectives > $O/a.txt 2> $O/a.err
echo "cached view:"; grep -E "^iteration" $O/a.txt | tr '\n' ' '; echo
CNMFE_SKIP_STITCH_ALLREDUCE=1 timeout 300 python scripts/host_timeline.py --patch 128 --iters 10 --force-collectives > $O/b.txt 2> $O/b.err
echo "no stitch all-reduce:"; grep -E "^iteration" $O/b.txt | tr '\n' ' '; echo
tail -3 $O/a.err
