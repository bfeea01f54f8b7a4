#!/bin/bash
# Round-3 record: bench lines of every configuration DESIGN.md section 4 quotes + the rocprofv3 kernel statistics and PMC passes of the bench command.
# On the GPU box: bash scripts/gpu_r3_profiles.sh <ver>   -> gpurun_out/r03/*_<ver>.json, gpurun_out/prof_r03<ver>/
ver=${1:-v1}; o=gpurun_out/r03; mkdir -p $o
python bench.py > $o/bench_c3_$ver.json 2> $o/bench_c3_$ver.err
X="--no-extras --no-cpu-baseline"
python bench.py $X --alg hals_thresh > $o/bench_c3_hals_thresh_$ver.json 2>/dev/null
python bench.py $X --alg nnls > $o/bench_c3_nnls_$ver.json 2>/dev/null
python bench.py $X --bg-ssub 2 > $o/bench_c3_bg_ssub2_$ver.json 2>/dev/null
python bench.py $X --deconv > $o/bench_c3_deconv_$ver.json 2>/dev/null
python bench.py $X --bg-ssub 2 --deconv > $o/bench_c3_demo_defaults_$ver.json 2>/dev/null
python bench.py $X --config c2 > $o/bench_c2_$ver.json 2>/dev/null
python bench.py $X --config c4 --steps 5 > $o/bench_c4_n1_$ver.json 2>/dev/null
python bench.py $X --warmup 0 --steps 5 > $o/bench_c3_warmup0_$ver.json 2>/dev/null
python bench.py $X --demo-sequence > $o/bench_c3_demo_sequence_$ver.json 2>/dev/null
bash scripts/profile_round.sh r03$ver > /dev/null 2>&1
bash scripts/gpu_gap.sh > /dev/null 2>&1
python scripts/host_timeline.py > $o/host_timeline_c3_$ver.txt 2>/dev/null
for f in $o/bench_*_$ver.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-48s %8.3f %-22s ms/step %s  kernel sum %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d.get("ms_per_step"), d.get("kernel_sum_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
