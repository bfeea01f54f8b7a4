#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_driver_like.json 2> gpurun_out/r5/bench_driver_like.err
echo "wall seconds: $SECONDS"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_driver_like.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['n_gpus'], d['steps'], d['warmup'], d['dtype'], d['vs_baseline'])
r=d['roofline']; print(r['kernel'], r['frac'], r['traffic'], r['traffic_source'][:60]); print(r['video_passes']); print(r['first_iteration_ms'], r['ms_per_step_incl_cold'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind'])
PY
