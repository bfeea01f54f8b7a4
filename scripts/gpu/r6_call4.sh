#!/bin/bash
mkdir -p gpurun_out
(
set -x
timeout 900 python scripts/solve_ab.py --cfg c3 --modes 0,1 --probes 0,16,32,64,24 --reps 2
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q -k "ring or packed or background" 2>&1 | tail -15
) > gpurun_out/r6_call4.log 2>&1
grep -v "^+\|amdgpu.ids" gpurun_out/r6_call4.log | tail -40
