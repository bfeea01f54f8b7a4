#!/bin/bash
# round 6, call 41: the round's final record (every bench configuration + rocprofv3 kernel statistics and PMC passes) on the final code
mkdir -p gpurun_out/r06
bash scripts/gpu_r6_profiles.sh v4 > gpurun_out/r06/record_run_v4.log 2>&1
cat gpurun_out/r06/record_v4.txt
