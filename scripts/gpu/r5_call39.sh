#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ssub_virtual.py tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_zconfigs.py -x -q -k "ssub or maps or consumer or demo or packed_iterations" > gpurun_out/r5/tests39.txt 2>&1; tail -n 3 gpurun_out/r5/tests39.txt
for i in 1 2; do timeout 300 python bench.py --bg-ssub 2 --no-cpu-baseline --no-extras > gpurun_out/r5/bench39_ssub2_$i.json 2> /dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/bench39_*.json')):
    d=json.load(open(f)); k=d['kernels_ms_per_step']; print(f, '%.3f ms/step  kernel sum %.3f  bg_win_proj %.3f  first it %.1f'%(d['ms_per_step'], d['kernel_sum_ms_per_step'], k['bg_win_proj'], d['first_iteration']['ms']))
PY
