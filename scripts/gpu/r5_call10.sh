#!/bin/bash
set -x
mkdir -p gpurun_out/r5; cd /root/repo
timeout 300 python scripts/gram_i8_check.py --cfg c3 > gpurun_out/r5/win_i8_check2.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench10_c3.json 2> gpurun_out/r5/bench10_c3.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --config c2 > gpurun_out/r5/bench10_c2.json 2> gpurun_out/r5/bench10_c2.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --config c4 --steps 4 > gpurun_out/r5/bench10_c4.json 2> gpurun_out/r5/bench10_c4.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --bg-ssub 2 > gpurun_out/r5/bench10_ssub2.json 2> gpurun_out/r5/bench10_ssub2.err
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5/tests10_full.txt 2>&1
grep -v "amdgpu\|^+" gpurun_out/r5/win_i8_check2.txt; tail -n 3 gpurun_out/r5/tests10_full.txt
