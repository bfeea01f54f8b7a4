#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sweep_dag.py tests/test_gpu_lanes.py -x -q 2>&1 | grep -a "passed\|failed\|rror\|assert" | tail -8
for l in 1 3; do
CNMFE_BENCH_LANES=$l timeout 280 python bench.py --config c4 --steps 6 --warmup 9 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c4 lanes', d['config'].get('lanes_per_rank'), round(d['ms_per_step'],2), d['first_iteration']['warmup_steps_ms'], d['kernel_calls_per_step'].get('spatial_hals_level'), d['kernels_ms_per_step'].get('temporal_hals_level'))"
done
for dag in 0 1; do
CNMFE_OPTS=sweep_dag=$dag timeout 280 python bench.py --deconv --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c3 deconv dag $dag', round(d['ms_per_step'],2), {k:v for k,v in d['kernels_ms_per_step'].items() if 'level' in k}, {k:v for k,v in d['kernel_calls_per_step'].items() if 'level' in k})"
CNMFE_OPTS=sweep_dag=$dag timeout 280 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c3 dag $dag', round(d['ms_per_step'],2), {k:v for k,v in d['kernels_ms_per_step'].items() if 'level' in k}, {k:v for k,v in d['kernel_calls_per_step'].items() if 'level' in k})"
done
