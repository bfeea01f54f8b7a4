#!/bin/bash
# round 4, GPU call 3: the new tests (K change, demo defaults, c3 full size), the whole suite, the default bench line, rocprofv3 kernel stats of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c3; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kchange.py tests/test_gpu_zconfigs.py -x -q -k "k_changes or demo_defaults or c3_full" > $O/test_new.txt 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt; tail -40 $O/test_new.txt >> $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -3 $O/smoke.txt >> $O/summary.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" >> $O/summary.txt
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err ); echo "rocprof rc=$?" >> $O/summary.txt
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_c3_kernel_stats.csv
rm -rf $O/prof
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/test_all.txt 2>&1; echo "all tests rc=$?" | tee -a $O/summary.txt; tail -30 $O/test_all.txt >> $O/summary.txt
python - <<'PY' >> gpurun_out/r4c3/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c3/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j.get("kernels_ms_per_step", {})
        print(f.split("bench_")[1][:-5], "ms/step %.2f" % j["ms_per_step"], "sum", j.get("kernel_sum_ms_per_step"), "r1", (j.get("roofline_r1") or {}).get("frac"), "roof", (j.get("roofline") or {}).get("kernel"), (j.get("roofline") or {}).get("frac"), "first", j["first_iteration"]["ms"], "c4", (j.get("c4_n1") or {}).get("ms_per_step"))
        print("   proj", {a: round(b["frac"], 3) for a, b in (j.get("roofline_projections") or {}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
