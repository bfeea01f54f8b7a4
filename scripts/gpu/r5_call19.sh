#!/bin/bash
set -x
mkdir -p gpurun_out/r5; cd /root/repo
for i in 1 2; do
timeout 300 python bench.py --bg-ssub 2 --no-cpu-baseline --no-extras > gpurun_out/r5/bench19_ssub2_virt_$i.json 2> gpurun_out/r5/bench19_ssub2_virt_$i.err
CNMFE_OPTS=ssub_virtual=0 timeout 300 python bench.py --bg-ssub 2 --no-cpu-baseline --no-extras > gpurun_out/r5/bench19_ssub2_swept_$i.json 2> /dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/bench19_*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable', e); continue
    k=d['kernels_ms_per_step']
    print(f, '%.3f ms/step  kernel sum %.3f'%(d['ms_per_step'], d['kernel_sum_ms_per_step']))
    print('    '+', '.join('%s %.2f'%(n,v) for n,v in sorted(k.items(), key=lambda x:-x[1])[:14]))
PY
tail -n 5 gpurun_out/r5/bench19_ssub2_virt_1.err
