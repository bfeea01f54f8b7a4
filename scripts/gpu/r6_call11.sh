#!/bin/bash
# round 6, call 11: first run of the ring solve out of cached inverses (ring_solve_inv.hpp) against the factorising kernel
mkdir -p gpurun_out/r06
for c in small edge c2 c3; do
  echo "== $c"; timeout 600 python scripts/probes/solve_inv/check_gpu.py --cfg $c 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06/solve_inv_call11.txt 2>&1
cat gpurun_out/r06/solve_inv_call11.txt
