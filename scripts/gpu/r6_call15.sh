#!/bin/bash
# execution lanes: the equality tests, then one rank's share and c4 on one GPU with 1 and 2 lanes
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_lanes.py -x -q 2>&1 | grep -a "passed\|failed\|rror\|assert\|Error" | tail -15
for l in 1 2; do
timeout 200 python scripts/rank_load.py --world 8 --steps 20 --lanes $l 2>&1 | grep -a "^rank" | cut -c1-200
done
for l in 1 2; do
CNMFE_BENCH_LANES=$l timeout 280 python bench.py --config c4 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c4 lanes', d['config'].get('lanes_per_rank'), d['ms_per_step'], 'kernel sum', d.get('kernel_sum_ms_per_step'))"
done
