#!/bin/bash
mkdir -p gpurun_out
(
set -x
timeout 900 python scripts/solve_ab.py --cfg c3 --modes 1 --probes 256 --reps 1
timeout 900 python scripts/solve_ab.py --cfg c2 --modes 1 --probes 256 --reps 1
) > gpurun_out/r6_call3.log 2>&1
grep -v "^+\|amdgpu.ids" gpurun_out/r6_call3.log | tail -40
