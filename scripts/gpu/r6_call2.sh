#!/bin/bash
# round 6, call 2: phase probes of the fp32 solve (3 waves/SIMD build), and the 2 waves/SIMD variant
mkdir -p gpurun_out
(
set -x
timeout 900 python scripts/solve_ab.py --cfg c3 --modes 1 --probes 0,16,32,64,24,48 --reps 2
CNMFE_LIB=cnmf_e_amd/variants/libcnmfe_w2.so timeout 900 python scripts/solve_ab.py --cfg c3 --modes 1 --probes 0,16,32,64 --reps 2
) > gpurun_out/r6_call2.log 2>&1
grep -v "^+\|amdgpu.ids" gpurun_out/r6_call2.log | tail -40
