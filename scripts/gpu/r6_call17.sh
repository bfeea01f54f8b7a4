#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_parity.py -x -q -k "lanes or method_level" 2>&1 | grep -a "passed\|failed\|rror" | tail -5
for l in 1 2; do
for fc in "" "--force-collectives"; do
timeout 200 python scripts/rank_load.py --world 8 --steps 30 --lanes $l $fc 2>&1 | grep -a "^rank" | cut -c1-260
done; done
for l in 1 2 3; do
CNMFE_BENCH_LANES=$l timeout 280 python bench.py --config c4 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c4 lanes', d['config'].get('lanes_per_rank'), round(d['ms_per_step'],2), 'kernel sum', d.get('kernel_sum_ms_per_step'))"
done
