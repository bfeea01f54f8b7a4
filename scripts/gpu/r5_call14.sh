#!/bin/bash
# round 5, GPU call 14: after the prune (split solve, fp32 / split-bf16 Gram modes; int8 digits also for the direct Gram): full GPU suite + headline
set -x
mkdir -p gpurun_out/r5; cd /root/repo
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench14_c3.json 2> gpurun_out/r5/bench14_c3.err
timeout 200 python scripts/incr_check.py > gpurun_out/r5/incr_check.txt 2>&1
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5/tests14_full.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5/tests14_full.txt | tail -n 3; tail -n 6 gpurun_out/r5/incr_check.txt
