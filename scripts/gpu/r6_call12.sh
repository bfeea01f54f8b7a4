#!/bin/bash
# round 6, call 12: the ring solve out of cached inverses against the factorising kernel (accuracy per fit, times; probe 16 = no ridge series)
mkdir -p gpurun_out/r06
for c in small edge c2 c3; do
  echo "== $c"; timeout 600 python scripts/probes/solve_inv/check_gpu.py --cfg $c 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06/solve_inv_call12.txt 2>&1
echo "== c3, no series (probe 16 + 512)" >> gpurun_out/r06/solve_inv_call12.txt
timeout 600 python scripts/probes/solve_inv/check_gpu.py --cfg c3 --probe 528 --modes 0,2 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/solve_inv_call12.txt
cat gpurun_out/r06/solve_inv_call12.txt
