#!/bin/bash
# runtime knobs of the process (not of the machine): polling instead of interrupts for host waits
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" "HSA_ENABLE_INTERRUPT=0"; do
echo "== $v"
env $v timeout 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c3', round(d['ms_per_step'],3))"
env $v timeout 200 python scripts/rank_load.py --world 8 --steps 40 --lanes 2 2>&1 | grep -a "^rank" | cut -c1-90
env $v timeout 200 python scripts/rank_load.py --world 8 --steps 40 --lanes 2 --force-collectives 2>&1 | grep -a "^rank" | cut -c1-110
done; done
