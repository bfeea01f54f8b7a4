#!/bin/bash
# Round-2 evidence run on the GPU box: rocprofv3 kernel stats + FETCH/WRITE PMC passes of the bench command, then the bench lines of the
# configurations DESIGN.md quotes.  Everything lands under gpurun_out/r02/ (copied into profiles/r02/ afterwards).
set -u
tag=${1:-v1}
o=gpurun_out/r02; mkdir -p $o
bash scripts/profile_round.sh r02_$tag
cp gpurun_out/prof_r02_$tag/kernel_stats.csv $o/bench_c3_kernel_stats_$tag.csv
cp gpurun_out/prof_r02_$tag/pmc_FETCH_SIZE.csv $o/bench_c3_pmc_FETCH_SIZE_$tag.csv
cp gpurun_out/prof_r02_$tag/pmc_WRITE_SIZE.csv $o/bench_c3_pmc_WRITE_SIZE_$tag.csv
cp gpurun_out/prof_r02_$tag/bench_under_rocprof.json $o/bench_c3_under_rocprof_$tag.json
python bench.py > $o/bench_c3_$tag.json 2> $o/bench_c3_$tag.err
python bench.py --config c4 --no-cpu-baseline > $o/bench_c4_n1_$tag.json 2>/dev/null
python bench.py --demo-sequence --no-cpu-baseline > $o/bench_c3_demo_sequence_$tag.json 2>/dev/null
python bench.py --warmup 0 --steps 5 --no-cpu-baseline > $o/bench_c3_warmup0_$tag.json 2>/dev/null
python bench.py --bg-ssub 2 --no-cpu-baseline > $o/bench_c3_bg_ssub2_$tag.json 2>/dev/null
python bench.py --deconv --no-cpu-baseline > $o/bench_c3_deconv_$tag.json 2>/dev/null
python bench.py --deconv --bg-ssub 2 --no-cpu-baseline > $o/bench_c3_demo_defaults_$tag.json 2>/dev/null
python bench.py --alg hals_thresh --no-cpu-baseline > $o/bench_c3_hals_thresh_$tag.json 2>/dev/null
python bench.py --alg nnls --no-cpu-baseline > $o/bench_c3_nnls_$tag.json 2>/dev/null
python bench.py --config c2 --no-cpu-baseline > $o/bench_c2_$tag.json 2>/dev/null
for f in $o/bench_*_$tag.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-46s %8.2f ms/step  value %.3f %s  kernel sum %s" % (sys.argv[1].split('/')[-1], d.get("ms_per_step", float('nan')), d.get("value", float('nan')), d.get("unit", ""), d.get("kernel_sum_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
