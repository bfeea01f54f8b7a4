#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/gpu/r6_call32.sh 2>&1 | grep -A6 "==== update_temporal" | cut -c1-150
for i in 1 2; do timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c3', round(d['ms_per_step'],2), 'first iteration', d['first_iteration']['ms'], d['first_iteration']['warmup_steps_ms'])"; done
