#!/bin/bash
# one rank's share with the collectives' host side: the round's gather (one collective, whole-A post-process) against the previous one (git stash of sources2d.py kept as sources2d_prev.py is NOT shipped: A/B by env)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 250 python scripts/rank_load.py --world 8 --steps 20 --force-collectives 2>&1 | grep -a "^rank" | cut -c1-220
done
timeout 250 python scripts/rank_load.py --world 8 --steps 20 2>&1 | grep -a "^rank" | cut -c1-220
CNMFE_BENCH_FORCE_COLLECTIVES=1 timeout 280 python bench.py --config c4 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 forced collectives', d['ms_per_step'])"
