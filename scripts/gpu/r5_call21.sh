#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ssub_virtual.py -x -q > gpurun_out/r5/tests21.txt 2>&1
timeout 300 python bench.py --bg-ssub 2 --no-cpu-baseline --no-extras > gpurun_out/r5/bench21_ssub2.json 2> /dev/null
CNMFE_OPTS=host_trace=1 timeout 300 python scripts/host_timeline.py --bg-ssub 2 > gpurun_out/r5/host_timeline_ssub2.txt 2>&1
tail -n 8 gpurun_out/r5/tests21.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench21_ssub2.json')); k=d['kernels_ms_per_step']
print('%.3f ms/step  kernel sum %.3f'%(d['ms_per_step'], d['kernel_sum_ms_per_step']))
print('    '+', '.join('%s %.2f'%(n,v) for n,v in sorted(k.items(), key=lambda x:-x[1])[:14]))
PY
