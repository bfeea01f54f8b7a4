#!/bin/bash
# round 6, call 22: the round's record on the final code (every bench configuration DESIGN.md quotes + rocprofv3 kernel statistics and PMC passes)
mkdir -p gpurun_out/r06
bash scripts/gpu_r6_profiles.sh v3 > gpurun_out/r06/record_run_v3.log 2>&1
cat gpurun_out/r06/record_v3.txt
