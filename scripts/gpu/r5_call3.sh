#!/bin/bash
# round 5, GPU call 3: shared diagonal steps (four pixels per workgroup), solve_variant 6..9
set -x
mkdir -p gpurun_out/r5
cd /root/repo
timeout 300 python scripts/solve_ab.py --cfg c3 --modes 0,6,7,8,9 --probes 0 --reps 3 > gpurun_out/r5/solve_ab3_c3.txt 2>&1
timeout 200 python scripts/solve_ab.py --cfg c3 --modes 6 --probes 2,4,8 --reps 1 > gpurun_out/r5/solve_ab3_c3_phases.txt 2>&1
CNMFE_OPTS=solve_variant=6 timeout 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q > gpurun_out/r5/tests3_variant6.txt 2>&1
CNMFE_OPTS=solve_variant=6 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench3_c3_var6.json 2> gpurun_out/r5/bench3_c3_var6.err
tail -n 9 gpurun_out/r5/solve_ab3_c3.txt; tail -n 4 gpurun_out/r5/tests3_variant6.txt
