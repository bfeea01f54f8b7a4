#!/bin/bash
# round 6, call 14: the matrix-vector product with DPP row sums and one barrier -- accuracy, then phase probes at c3
mkdir -p gpurun_out/r06
o=gpurun_out/r06/solve_inv_call14.txt
for c in small c3; do
  echo "== $c"; timeout 600 python scripts/probes/solve_inv/check_gpu.py --cfg $c --modes 0,2,1 2>&1 | grep -v amdgpu.ids | grep -v "^   pixel"
done > $o 2>&1
for pr in 544 576 640 528; do
  echo "== c3 probe $pr (32: set-up + loads, 64: + matrix pipe, 128: + small inversion, 16: no ridge series; +512 statistics)" >> $o
  timeout 600 python scripts/probes/solve_inv/check_gpu.py --cfg c3 --probe $pr --modes 2 2>&1 | grep -v amdgpu.ids | grep "mode 2 fit [01]" >> $o
done
cat $o
