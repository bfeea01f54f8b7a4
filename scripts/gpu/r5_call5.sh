#!/bin/bash
# round 5, GPU call 5: the driver's bench command with the in-run PMC traffic; a slice of the GPU suite
set -x
mkdir -p gpurun_out/r5; cd /root/repo
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench5_default.json 2> gpurun_out/r5/bench5_default.err ) 2> gpurun_out/r5/bench5_time.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_kchange.py -x -q > gpurun_out/r5/tests5.txt 2>&1
tail -3 gpurun_out/r5/tests5.txt; cat gpurun_out/r5/bench5_time.txt
