#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
for i in 1 2 3 4 5; do
CNMFE_OPTS=host_trace=2 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/r5/stall_$i.json 2> gpurun_out/r5/stall_$i.err
python - <<PY
import json
d=json.load(open('gpurun_out/r5/stall_$i.json')); print($i, d['first_iteration']['ms'], d['first_iteration']['kernel_sum_ms'], d['ms_per_step'])
PY
done
