#!/bin/bash
set -x
mkdir -p gpurun_out/r5; cd /root/repo
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench17_base_$i.json 2> /dev/null
CNMFE_LIB=/root/repo/cnmf_e_amd/variants/libcnmfe_nt.so timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench17_nt_$i.json 2> /dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/bench17_*.json')):
    d=json.load(open(f)); k=d['kernels_ms_per_step']
    print(f, '%.3f ms/step  solve %.3f projB %.3f win %.3f'%(d['ms_per_step'],k['bg_ring_solve'],k['temporal_proj_B'],k['bg_win_proj']))
PY
