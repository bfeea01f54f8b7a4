#!/bin/bash
# round 4, GPU call 8: the ring solve in two halves (deferred half under the host's turnaround), T = 50000 fixes; targeted tests, bench A/B, then the parity suites
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c8; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_virtual.py tests/test_gpu_edges.py -x -q -k "deferred_half or get_sn_of" --durations=5 > $O/test_new.txt 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt; tail -15 $O/test_new.txt >> $O/summary.txt
for v in defer1 defer0; do
  CNMFE_OPTS=solve_defer=${v#defer} CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_$v.json 2> $O/bench_$v.err
done
CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --config c2 > $O/bench_c2.json 2> /dev/null
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_virtual.py tests/test_gpu_kchange.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -x -q --deselect tests/test_gpu_edges.py::test_bench_eight_ranks_full_size_c4_on_one_device > $O/test_most.txt 2>&1; echo "most tests rc=$?" | tee -a $O/summary.txt; tail -8 $O/test_most.txt >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_zconfigs.py -x -q -k "m128_hals or m128_2x2 or demo_defaults or c3_full" > $O/test_z.txt 2>&1; echo "z tests rc=$?" | tee -a $O/summary.txt; tail -8 $O/test_z.txt >> $O/summary.txt
python - <<'PY' >> gpurun_out/r4c8/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c8/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j["kernels_ms_per_step"]
        print(f.split("bench_")[1][:-5], "ms/step %.2f" % j["ms_per_step"], "sum", j["kernel_sum_ms_per_step"], "solve", k.get("bg_ring_solve"), j["kernel_calls_per_step"].get("bg_ring_solve"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
