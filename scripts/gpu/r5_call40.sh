#!/bin/bash
mkdir -p gpurun_out/r05; cd /root/repo; o=gpurun_out/r05
export CNMFE_BENCH_R1=0
X="--no-extras --no-cpu-baseline"
python bench.py $X --deconv > $o/bench_c3_deconv_v3.json 2>/dev/null
python bench.py $X --bg-ssub 2 --deconv --alg hals_thresh > $o/bench_c3_demo_defaults_v3.json 2>/dev/null
python bench.py $X --config c5shard --steps 5 > $o/bench_c5shard_v3.json 2>/dev/null
python bench.py $X --demo-sequence > $o/bench_c3_demo_sequence_v3.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05/bench_*_v3.json')):
    d=json.load(open(f)); print('%-40s %s %s  ms/step %s  kernel sum %s'%(f.split('/')[-1], round(d['value'],3), d['unit'][:12], d.get('ms_per_step'), d.get('kernel_sum_ms_per_step')))
PY
