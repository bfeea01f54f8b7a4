#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
for i in 1 2 3 4; do
CNMFE_OPTS=host_trace=2 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/r5/stallc_$i.json 2> gpurun_out/r5/stallc_$i.err
python - <<PY
import json,re
d=json.load(open('gpurun_out/r5/stallc_$i.json')); print($i, round(d['first_iteration']['ms'],1), d['first_iteration']['kernel_sum_ms'], round(d['ms_per_step'],2))
n=0
for l in open('gpurun_out/r5/stallc_$i.err'):
    m=re.search(r'\s([0-9.]+) ms \(at', l)
    if (m and float(m.group(1))>8) or ('launch ' in l and 'flush' in l):
        print('   ', l.strip()[:150]); n+=1
        if n>12: break
PY
done
