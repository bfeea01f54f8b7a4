#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ssub_virtual.py tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_zconfigs.py -x -q -k "ssub or maps or consumer or demo or packed_iterations" > gpurun_out/r5/tests32.txt 2>&1; tail -n 3 gpurun_out/r5/tests32.txt
for i in 1 2 3; do timeout 300 python bench.py --bg-ssub 2 --no-cpu-baseline --no-extras > gpurun_out/r5/bench32_ssub2_$i.json 2> /dev/null; done
CNMFE_OPTS=host_trace=1 timeout 300 python scripts/host_timeline.py --bg-ssub 2 2>&1 | grep "down(A_prev)\|last iteration" | tail -3
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/bench32_*.json')):
    d=json.load(open(f)); print(f, '%.3f ms/step  kernel sum %.3f'%(d['ms_per_step'], d['kernel_sum_ms_per_step']))
PY
