#!/bin/bash
# record v5 (after the lanes / hand-over work): the GPU suite, the default bench line (c3 + its c4_n1 child), c4 behind a group of one, c5shard, one rank's share
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_v7_full.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_v7_full.txt
grep -a "passed\|failed\|pytest rc" gpurun_out/gputest_v7_full.txt > gpurun_out/gputest_v7.txt
timeout 400 python bench.py 2> gpurun_out/bench_c3_v5.err | grep -a "^{" > gpurun_out/bench_c3_v5.json
CNMFE_BENCH_FORCE_COLLECTIVES=1 timeout 280 python bench.py --config c4 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | grep -a "^{" > gpurun_out/bench_c4_forced_collectives_v5.json
timeout 280 python bench.py --config c4 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | grep -a "^{" > gpurun_out/bench_c4_n1_v5.json
timeout 280 python bench.py --config c5shard --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep -a "^{" > gpurun_out/bench_c5shard_v5.json
timeout 280 python bench.py --config c4 --deconv --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep -a "^{" > gpurun_out/bench_c4_deconv_v5.json
( for l in 1 2; do for fc in "" "--force-collectives"; do timeout 200 python scripts/rank_load.py --world 8 --steps 30 --lanes $l $fc 2>&1 | grep -a "^rank\|^{" | cut -c1-600; done; done ) > gpurun_out/rank_load_v5.txt
cat gpurun_out/gputest_v7.txt
for f in bench_c3_v5 bench_c4_forced_collectives_v5 bench_c4_n1_v5 bench_c5shard_v5 bench_c4_deconv_v5; do python -c "
import json,sys
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],2), d['config'].get('lanes_per_rank'), (d.get('c4_n1') or {}).get('ms_per_step'))"; done
grep -a "^rank" gpurun_out/rank_load_v5.txt | cut -c1-120
