#!/bin/bash
# round 5, GPU call 1: A/B of the ring-solve variants (option solve_variant) at the headline size, phase probes, parity tests and the bench line with variant 2
set -x
mkdir -p gpurun_out/r5
cd /root/repo
timeout 600 python scripts/solve_ab.py --cfg c3 --modes 0,1,2 --probes 0 --reps 3 > gpurun_out/r5/solve_ab_c3.txt 2>&1
timeout 300 python scripts/solve_ab.py --cfg c3 --modes 0,2 --probes 2,4,8 --reps 1 > gpurun_out/r5/solve_ab_c3_phases.txt 2>&1
CNMFE_OPTS=solve_variant=2 timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q > gpurun_out/r5/tests_variant2.txt 2>&1
CNMFE_OPTS=solve_variant=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench_c3_var0.json 2> gpurun_out/r5/bench_c3_var0.err
CNMFE_OPTS=solve_variant=2 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench_c3_var2.json 2> gpurun_out/r5/bench_c3_var2.err
CNMFE_OPTS=solve_variant=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench_c3_var1.json 2> gpurun_out/r5/bench_c3_var1.err
tail -5 gpurun_out/r5/solve_ab_c3.txt gpurun_out/r5/tests_variant2.txt
