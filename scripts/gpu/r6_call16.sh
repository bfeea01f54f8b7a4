#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 1 2 3 4; do
CNMFE_BENCH_LANES=$l timeout 280 python bench.py --config c4 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c4 lanes', d['config'].get('lanes_per_rank'), round(d['ms_per_step'],2), 'kernel sum', d.get('kernel_sum_ms_per_step'))"
done
for l in 1 2 4; do
CNMFE_BENCH_LANES=$l timeout 280 python bench.py --config c5shard --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c5shard lanes', d['config'].get('lanes_per_rank'), round(d['ms_per_step'],2), 'kernel sum', d.get('kernel_sum_ms_per_step'))"
done
