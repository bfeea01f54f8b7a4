#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_int8.py tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_edges.py tests/test_gpu_kchange.py tests/test_golden.py -x -q > gpurun_out/r5/tests35.txt 2>&1; tail -n 6 gpurun_out/r5/tests35.txt
