#!/bin/bash
mkdir -p gpurun_out/r05; cd /root/repo; o=gpurun_out/r05
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_c3_v3.json 2> $o/bench_c3_v3.err
export CNMFE_BENCH_R1=0
X="--no-extras --no-cpu-baseline"
python bench.py $X --bg-ssub 2 > $o/bench_c3_bg_ssub2_v3.json 2>/dev/null
python bench.py $X --config c4 --steps 5 > $o/bench_c4_n1_v3.json 2>/dev/null
python bench.py $X --config c2 > $o/bench_c2_v3.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05/bench_*_v3.json')):
    d=json.load(open(f)); k=d['kernels_ms_per_step']
    print('%-34s %.2f iter/s  %.3f ms/step  kernel sum %.3f  top: %s'%(f.split('/')[-1], d['value'], d['ms_per_step'], d['kernel_sum_ms_per_step'], ', '.join('%s %.2f'%(n,v) for n,v in sorted(k.items(), key=lambda x:-x[1])[:4])))
d=json.load(open('gpurun_out/r05/bench_c3_v3.json')); r=d['roofline']; print(r['frac'], r['video_passes'], r['first_iteration_ms'], r['traffic_source'][:40])
PY
