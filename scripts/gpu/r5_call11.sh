#!/bin/bash
set -x
mkdir -p gpurun_out/r5; cd /root/repo
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench11_c3.json 2> gpurun_out/r5/bench11_c3.err
CNMFE_OPTS=proj_tiled=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench11_c3_notile.json 2> gpurun_out/r5/bench11_c3_notile.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --config c4 --steps 4 > gpurun_out/r5/bench11_c4.json 2> gpurun_out/r5/bench11_c4.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_virtual.py tests/test_gpu_kchange.py -x -q > gpurun_out/r5/tests11.txt 2>&1
tail -n 3 gpurun_out/r5/tests11.txt
