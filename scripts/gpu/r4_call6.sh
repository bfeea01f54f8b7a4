#!/bin/bash
# round 4, GPU call 6: the eight-rank c4 run on one device, configs[4] whole on one GPU, cold start with the code objects preloaded, the full suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c6; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_zconfigs.py -x -q -k "eight_ranks or c5_whole" --durations=5 > $O/test_new.txt 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt; tail -25 $O/test_new.txt >> $O/summary.txt
for rep in 1 2; do
  CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 2 > $O/cold_$rep.json 2> $O/cold_$rep.err
done
CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --demo-sequence > $O/demo_seq.json 2> /dev/null
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $O/test_all.txt 2>&1; echo "all tests rc=$?" | tee -a $O/summary.txt; tail -20 $O/test_all.txt >> $O/summary.txt
python - <<'PY' >> gpurun_out/r4c6/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c6/cold_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["first_iteration"]["warmup_steps_ms"], "steady", round(j["ms_per_step"], 2))
    except Exception as e:
        print(f, "ERR", e)
try:
    j = json.loads(open("gpurun_out/r4c6/demo_seq.json").read().strip().splitlines()[-1]); print("demo_seq", j["value"])
except Exception as e:
    print("demo ERR", e)
PY
cat $O/summary.txt
