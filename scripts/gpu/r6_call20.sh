#!/bin/bash
# round 6, call 20: three digit planes in the temporal projection -- parity tests, then the bench with 3 and 4 planes
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_int8.py tests/test_gpu_virtual.py tests/test_gpu_solve_r6.py -x -q 2>&1 | tail -5
X="--no-extras --no-cpu-baseline --steps 20 --warmup 3"
for pl in 3 4 3 4; do
  CNMFE_OPTS=proj_i8_planes=$pl python bench.py $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('planes $pl: %.3f ms/step  proj_B %.3f  win_proj %.3f  solve %.3f  kernel sum %.3f' % (d['ms_per_step'], k['temporal_proj_B'], k['bg_win_proj'], k['bg_ring_solve'], d['kernel_sum_ms_per_step']))"
done
