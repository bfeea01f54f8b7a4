#!/bin/bash
set -x
mkdir -p gpurun_out/r5; cd /root/repo
timeout 300 python scripts/ssub_virtual_check.py --cfg small > gpurun_out/r5/ssubv_small.txt 2>&1
timeout 300 python scripts/ssub_virtual_check.py --cfg odd --pdims 40,33 > gpurun_out/r5/ssubv_odd.txt 2>&1
timeout 300 python scripts/ssub_virtual_check.py --cfg mid > gpurun_out/r5/ssubv_mid.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ssub" > gpurun_out/r5/tests18.txt 2>&1
tail -n 12 gpurun_out/r5/ssubv_small.txt gpurun_out/r5/ssubv_odd.txt gpurun_out/r5/ssubv_mid.txt; tail -n 15 gpurun_out/r5/tests18.txt
