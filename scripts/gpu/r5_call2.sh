#!/bin/bash
# round 5, GPU call 2: the ring-solve variants after the DPP-hazard fix, with the rank-2 corrections on the matrix pipe
set -x
mkdir -p gpurun_out/r5
cd /root/repo
timeout 600 python scripts/solve_ab.py --cfg c3 --modes 0,1,2,3,4,5 --probes 0 --reps 3 > gpurun_out/r5/solve_ab2_c3.txt 2>&1
timeout 300 python scripts/solve_ab.py --cfg c3 --modes 0,4 --probes 2,4,8 --reps 1 > gpurun_out/r5/solve_ab2_c3_phases.txt 2>&1
CNMFE_OPTS=solve_variant=4 timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q > gpurun_out/r5/tests2_variant4.txt 2>&1
CNMFE_OPTS=solve_variant=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench2_c3_var0.json 2> gpurun_out/r5/bench2_c3_var0.err
CNMFE_OPTS=solve_variant=4 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench2_c3_var4.json 2> gpurun_out/r5/bench2_c3_var4.err
tail -n 9 gpurun_out/r5/solve_ab2_c3.txt; tail -n 4 gpurun_out/r5/tests2_variant4.txt
