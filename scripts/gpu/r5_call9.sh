#!/bin/bash
# round 5, GPU call 9: the window projection on the int8 pipe (win_proj_i8.hpp)
set -x
mkdir -p gpurun_out/r5; cd /root/repo
for c in small r18 c2 c3; do timeout 300 python scripts/gram_i8_check.py --cfg $c; done > gpurun_out/r5/win_i8_check.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_virtual.py tests/test_gpu_kchange.py -x -q > gpurun_out/r5/tests9.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench9_c3.json 2> gpurun_out/r5/bench9_c3.err
CNMFE_OPTS=win_i8=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench9_c3_nowin.json 2> gpurun_out/r5/bench9_c3_nowin.err
grep -v "amdgpu\|^+" gpurun_out/r5/win_i8_check.txt; tail -n 3 gpurun_out/r5/tests9.txt
