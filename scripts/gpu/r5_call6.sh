#!/bin/bash
# round 5, GPU call 6: the int8-digit covariance table (gram_i8.hpp)
set -x
mkdir -p gpurun_out/r5; cd /root/repo
for c in small r18 c2 c3; do timeout 300 python scripts/gram_i8_check.py --cfg $c; done > gpurun_out/r5/gram_i8_check.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -k "incremental_gram or outlier or stride" > gpurun_out/r5/tests6a.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py -x -q > gpurun_out/r5/tests6b.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench6_c3.json 2> gpurun_out/r5/bench6_c3.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --demo-sequence > gpurun_out/r5/bench6_demo_seq.json 2> gpurun_out/r5/bench6_demo_seq.err
grep -v amdgpu gpurun_out/r5/gram_i8_check.txt; tail -3 gpurun_out/r5/tests6a.txt gpurun_out/r5/tests6b.txt
