#!/bin/bash
# round 6, call 1: the fp32 + refinement ring solve -- correctness on the small case, then timing at c3
mkdir -p gpurun_out
(
set -x
timeout 600 python scripts/solve_ab.py --cfg small --modes 0,1 --probes 0 --reps 1
timeout 900 python scripts/solve_ab.py --cfg c3 --modes 0,1 --probes 0,16,2 --reps 3
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py -x -q -k "ring or packed or background" 2>&1 | tail -15
) > gpurun_out/r6_call1.log 2>&1
tail -40 gpurun_out/r6_call1.log
