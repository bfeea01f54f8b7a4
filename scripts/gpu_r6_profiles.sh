#!/bin/bash
# Round-6 record: bench lines of every configuration DESIGN.md section 4 quotes + the rocprofv3 kernel statistics and PMC passes of the bench command.
# On the GPU box: bash scripts/gpu_r5_profiles.sh <ver>   -> gpurun_out/r06/*_<ver>.json, gpurun_out/prof_r06<ver>/
cd "$GRAFT_REPO_ROOT" || exit 1
ver=${1:-v1}; o=gpurun_out/r06; mkdir -p $o
export PYTHONUNBUFFERED=1
python bench.py > $o/bench_c3_$ver.json 2> $o/bench_c3_$ver.err
X="--no-extras --no-cpu-baseline"
export CNMFE_BENCH_R1=0
python bench.py $X --alg hals_thresh > $o/bench_c3_hals_thresh_$ver.json 2>/dev/null
python bench.py $X --alg nnls > $o/bench_c3_nnls_$ver.json 2>/dev/null
python bench.py $X --bg-ssub 2 > $o/bench_c3_bg_ssub2_$ver.json 2>/dev/null
CNMFE_OPTS=ssub_virtual=0 python bench.py $X --bg-ssub 2 > $o/bench_c3_bg_ssub2_swept_$ver.json 2>/dev/null
python bench.py $X --deconv > $o/bench_c3_deconv_$ver.json 2>/dev/null
python bench.py $X --bg-ssub 2 --deconv --alg hals_thresh > $o/bench_c3_demo_defaults_$ver.json 2>/dev/null
python bench.py $X --config c2 > $o/bench_c2_$ver.json 2>/dev/null
python bench.py $X --config c4 --steps 10 --warmup 5 > $o/bench_c4_n1_$ver.json 2>/dev/null
python bench.py $X --config c5shard --steps 5 --warmup 5 > $o/bench_c5shard_$ver.json 2>/dev/null
python bench.py $X --config c5shard --steps 5 --warmup 5 --deconv > $o/bench_c5shard_deconv_$ver.json 2>/dev/null
python bench.py $X --warmup 0 --steps 5 > $o/bench_c3_warmup0_$ver.json 2>/dev/null
python bench.py $X --demo-sequence > $o/bench_c3_demo_sequence_$ver.json 2>/dev/null
CNMFE_OPTS=r1_virtual=0 python bench.py $X > $o/bench_c3_swept_$ver.json 2>/dev/null
# round 6's switches, one at a time against the default line: the staged ring solve, the temporal projection's digit planes, the ring solve out of cached inverses
CNMFE_OPTS=solve_staged=0 python bench.py $X > $o/bench_c3_ab_solve_staged_off_$ver.json 2>/dev/null
CNMFE_OPTS=proj_i8_planes=4 python bench.py $X > $o/bench_c3_ab_proj_planes4_$ver.json 2>/dev/null
CNMFE_OPTS=solve_inv=1 python bench.py $X > $o/bench_c3_ab_solve_inv_on_$ver.json 2>/dev/null
python scripts/probes/solve_inv/bench_loop.py --cfg c3 --steps 12 --mode 1 2>&1 | grep -v amdgpu.ids > $o/solve_inv_bench_loop_$ver.txt
# the round's last switches: the sweeps as one dependency graph (sweep_dag), the execution lanes
CNMFE_OPTS=sweep_dag=0 python bench.py $X --deconv > $o/bench_c3_ab_deconv_sweep_dag_off_$ver.json 2>/dev/null
CNMFE_BENCH_LANES=1 python bench.py $X --config c4 --steps 10 --warmup 5 > $o/bench_c4_n1_ab_one_lane_$ver.json 2>/dev/null
CNMFE_BENCH_LANES=1 python bench.py $X --config c5shard --steps 5 --warmup 5 > $o/bench_c5shard_ab_one_lane_$ver.json 2>/dev/null
( for l in 1 2; do for fc in "" "--force-collectives"; do python scripts/rank_load.py --world 8 --steps 30 --lanes $l $fc 2>&1 | grep -a "^rank\|^{" | cut -c1-700; done; done ) > $o/rank_load_$ver.txt 2>&1
CNMFE_BENCH_FORCE_COLLECTIVES=1 python bench.py $X --config c4 --steps 10 --warmup 4 > $o/bench_c4_forced_collectives_$ver.json 2>/dev/null
unset CNMFE_BENCH_R1
bash scripts/profile_round.sh r06$ver > /dev/null 2>&1
for f in $o/bench_*_$ver.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-48s %8.3f %-22s ms/step %s  kernel sum %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d.get("ms_per_step"), d.get("kernel_sum_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done | tee $o/record_$ver.txt
bash scripts/pmc_round5.sh > /dev/null 2>&1; cp gpurun_out/pmc_r05/summary.txt $o/pmc_pipes_$ver.txt 2>/dev/null
tail -5 $o/rank_load_$ver.txt
