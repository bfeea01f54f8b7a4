#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ssub_virtual.py tests/test_gpu_parity.py tests/test_gpu_packed.py -x -q -k "ssub or packed_iterations or maps or consumer" > gpurun_out/r5/tests22.txt 2>&1
for i in 1 2; do timeout 300 python bench.py --bg-ssub 2 --no-cpu-baseline --no-extras > gpurun_out/r5/bench22_ssub2_$i.json 2> /dev/null; done
CNMFE_OPTS=host_trace=1 timeout 300 python scripts/host_timeline.py --bg-ssub 2 > gpurun_out/r5/host_timeline_ssub2_b.txt 2>&1
tail -n 4 gpurun_out/r5/tests22.txt
python - <<'PY'
import json
for i in (1,2):
    d=json.load(open('gpurun_out/r5/bench22_ssub2_%d.json'%i)); k=d['kernels_ms_per_step']
    print('%.3f ms/step  kernel sum %.3f'%(d['ms_per_step'], d['kernel_sum_ms_per_step']))
PY
tail -n 12 gpurun_out/r5/host_timeline_ssub2_b.txt
