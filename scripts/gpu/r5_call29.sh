#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 600 python -m pytest tests/test_gpu_zconfigs.py -x -q -k "c5_whole or c4_sixteen" > gpurun_out/r5/tests29a.txt 2>&1; tail -n 3 gpurun_out/r5/tests29a.txt
./scripts/probes/rsq/rsq_probe > gpurun_out/r5/rsq_probe.txt 2>&1; cat gpurun_out/r5/rsq_probe.txt
for i in 1 2; do for v in base nr1 sel both; do
  if [ $v = base ]; then unset CNMFE_LIB; else export CNMFE_LIB=/root/repo/cnmf_e_amd/variants/libcnmfe_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench29_${v}_$i.json 2> /dev/null
done; done
unset CNMFE_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/bench29_*.json')):
    d=json.load(open(f)); k=d['kernels_ms_per_step']
    print('%-44s %.3f ms/step  solve %.3f'%(f, d['ms_per_step'], k['bg_ring_solve']))
PY
CNMFE_LIB=/root/repo/cnmf_e_amd/variants/libcnmfe_both.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_int8.py tests/test_gpu_virtual.py -x -q > gpurun_out/r5/tests29b.txt 2>&1; tail -n 4 gpurun_out/r5/tests29b.txt
