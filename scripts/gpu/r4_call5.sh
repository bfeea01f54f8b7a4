#!/bin/bash
# round 4, GPU call 5: cold start -- the first iteration in fresh processes with and without the reserved buffers; c3 full-size test
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c5; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2 3; do
  for pre in 1 0; do
    CNMFE_OPTS=prealloc=$pre,host_trace=$([ $rep = 1 ] && echo 1 || echo 0) CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 2 > $O/cold_pre${pre}_$rep.json 2> $O/cold_pre${pre}_$rep.err
  done
done
CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --demo-sequence > $O/demo_seq_1.json 2> /dev/null
CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --demo-sequence > $O/demo_seq_2.json 2> /dev/null
python - <<'PY' > $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c5/cold_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["first_iteration"]["warmup_steps_ms"], j["first_iteration"]["one_off_kernels_ms"], "steady", round(j["ms_per_step"], 2))
    except Exception as e:
        print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r4c5/demo_seq_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], j["value"], {k: v for k, v in list(j["kernels_ms_total"].items())[:8]})
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_gpu_zconfigs.py -x -q -k "c3_full" > $O/test_c3.txt 2>&1; echo "c3 test rc=$?" >> $O/summary.txt; tail -5 $O/test_c3.txt >> $O/summary.txt
cat $O/summary.txt
