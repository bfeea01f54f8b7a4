#!/bin/bash
# round 6, call 10: the whole GPU suite, then the round's record (every bench configuration DESIGN.md quotes + rocprofv3 kernel statistics and PMC passes)
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r06/gputest_v2.txt
bash scripts/gpu_r6_profiles.sh v2 > gpurun_out/r06/record_run_v2.log 2>&1
cat gpurun_out/r06/gputest_v2.txt; cat gpurun_out/r06/record_v2.txt
