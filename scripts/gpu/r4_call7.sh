#!/bin/bash
# round 4, GPU call 7: long recordings (T = 50000), proj_B tuning variants, first-iteration kernel table, a kernel trace for the gap analysis
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c7; mkdir -p $O
export PYTHONUNBUFFERED=1
V=$PWD/cnmf_e_amd/variants
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py -x -q -k "long or get_sn or deconv" --durations=5 > $O/test_long.txt 2>&1; echo "long tests rc=$?" | tee -a $O/summary.txt; tail -15 $O/test_long.txt >> $O/summary.txt
for v in default ah16 ah4 wg4096 wg1024 ah16wg4096; do
  L=""; [ $v != default ] && L=$V/libcnmfe_$v.so
  CNMFE_LIB=$L CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 2 > $O/bench_$v.json 2> $O/bench_$v.err
done
( cd /tmp && export TMPDIR=/tmp && CNMFE_BENCH_R1=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/gap_analysis.py "$f" seq > $O/gap_analysis_c3.txt 2>&1
rm -rf $O/trace
python - <<'PY' >> gpurun_out/r4c7/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c7/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j["kernels_ms_per_step"]
        print(f.split("bench_")[1][:-5], "ms/step %.2f" % j["ms_per_step"], "proj_B", k.get("temporal_proj_B"), "reduce", k.get("temporal_reduce_B"), "first", j["first_iteration"]["warmup_steps_ms"])
        if "default" in f: print("   first-step kernels", j["first_iteration"]["kernels_ms"], j["first_iteration"]["kernel_sum_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt; head -60 $O/gap_analysis_c3.txt
