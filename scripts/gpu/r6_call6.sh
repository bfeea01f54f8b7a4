#!/bin/bash
# round 6, call 6: the advisor fixes + the lifted ring cap under the GPU tests they touch, the bench line, the round's first rocprof record
mkdir -p gpurun_out/r06
(
set -x
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_ssub_virtual.py tests/test_gpu_int8.py tests/test_gpu_packed.py -x -q -m gpu 2>&1 | tail -8
python bench.py > gpurun_out/r06/bench_c3_v1.json 2> gpurun_out/r06/bench_c3_v1.err
tail -c 1500 gpurun_out/r06/bench_c3_v1.json | head -c 600
bash scripts/profile_round.sh r06v1
) > gpurun_out/r6_call6.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r6_call6.log | tail -30
