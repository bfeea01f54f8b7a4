#!/bin/bash
mkdir -p gpurun_out
(
set -x
timeout 900 python scripts/solve_ab.py --cfg c3 --modes 1 --probes 0,16,1024,1536,512 --reps 2
) > gpurun_out/r6_call5.log 2>&1
grep -v "^+\|amdgpu.ids" gpurun_out/r6_call5.log | tail -40
