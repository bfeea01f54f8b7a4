#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_virtual.py tests/test_gpu_int8.py tests/test_gpu_parity.py -x -q > gpurun_out/r5/tests34.txt 2>&1; tail -n 6 gpurun_out/r5/tests34.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench34_i8_$i.json 2> gpurun_out/r5/bench34_i8_$i.err
CNMFE_OPTS=proj_i8=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench34_f64_$i.json 2> /dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/bench34_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels_ms_per_step']
        print('%-40s %.3f ms/step  proj_B %.3f  panel_dig %.3f  first it %.1f'%(f, d['ms_per_step'], k.get('temporal_proj_B',0), k.get('temporal_panel_dig',0), d['first_iteration']['ms']))
    except Exception as e: print(f, 'FAILED', e)
PY
tail -n 3 gpurun_out/r5/bench34_i8_1.err
