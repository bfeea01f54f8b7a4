#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c39; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
timeout 600 python -m pytest tests/test_gpu_virtual.py tests/test_gpu_parity.py tests/test_gpu_kchange.py -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt
CNMFE_OPTS=host_trace=1 timeout 300 python scripts/host_timeline.py --iters 4 > $O/ht.txt 2> $O/ht.err
grep "vproj" $O/ht.err | tail -4
X="--no-extras --no-cpu-baseline"
for i in 1 2; do timeout 300 python bench.py $X > $O/c3_$i.json 2> $O/c3_$i.err; done
timeout 300 python bench.py $X --config c4 --steps 5 > $O/c4.json 2> $O/c4.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c39/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], "kernel sum", j["kernel_sum_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
