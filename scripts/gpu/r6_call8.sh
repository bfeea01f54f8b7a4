#!/bin/bash
mkdir -p gpurun_out/r06
(
timeout 2400 python -m pytest tests -x -q -m gpu -k "c5 or stride or incremental or int8 or virtual or packed" 2>&1 | tail -6
python bench.py --no-extras --no-cpu-baseline --config c5shard --steps 5 > gpurun_out/r06/bench_c5shard_v1.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_c5shard_v1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('kernel_sum_ms_per_step'))
for k,v in list(d['kernels_ms_per_step'].items())[:14]: print("  %-30s %.3f"%(k,v))
PY
) > gpurun_out/r6_call8.log 2>&1
grep -v amdgpu gpurun_out/r6_call8.log | tail -30
