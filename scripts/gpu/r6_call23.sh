#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 3 1; do for w in 1 2 3 4 6; do
CNMFE_BENCH_LANES=$l timeout 280 python bench.py --config c4 --steps 3 --warmup $w --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c4 lanes $l warmup $w:', round(d['ms_per_step'],2), [round(x,1) for x in d['first_iteration']['warmup_steps_ms']])"
done; done
