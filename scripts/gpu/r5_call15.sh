#!/bin/bash
set -x
mkdir -p gpurun_out/r5; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_int8.py -x -q > gpurun_out/r5/tests15.txt 2>&1
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench15_c3_$i.json 2> /dev/null; done
CNMFE_OPTS=proj_tiled=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench15_c3_notile.json 2> /dev/null
tail -n 5 gpurun_out/r5/tests15.txt
