#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_int8.py -x -q 2>&1 | grep -a "passed\|failed\|assert" | tail -4
for rep in 1 2; do for pl in 4 3; do
CNMFE_OPTS=win_i8_planes=$pl timeout 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c3 win_i8_planes=$pl:', round(d['ms_per_step'],3), 'bg_win_proj', d['kernels_ms_per_step'].get('bg_win_proj'))"
done; done
