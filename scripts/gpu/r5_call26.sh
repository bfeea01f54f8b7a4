#!/bin/bash
cd /root/repo; o=gpurun_out/r05; mkdir -p $o
python bench.py > $o/bench_c3_v2.json 2> $o/bench_c3_v2.err
export CNMFE_BENCH_R1=0
for i in a b c; do python bench.py --no-extras --no-cpu-baseline --demo-sequence > $o/bench_c3_demo_sequence_v2$i.json 2>/dev/null; done
python bench.py --no-extras --no-cpu-baseline --bg-ssub 2 --demo-sequence > $o/bench_c3_demo_sequence_ssub2_v2.json 2>/dev/null
python - <<'PY'
import json,glob
d=json.load(open('gpurun_out/r05/bench_c3_v2.json')); print(d['value'], d['ms_per_step'], d['first_iteration']['ms'], d['roofline']['video_passes'])
for f in sorted(glob.glob('gpurun_out/r05/bench_c3_demo_sequence*_v2*.json')):
    try: print(f, json.load(open(f))['value'])
    except Exception as e: print(f, e)
PY
