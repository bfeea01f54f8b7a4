#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c21; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
X="--no-extras --no-cpu-baseline --steps 5"
timeout 300 python bench.py $X --config c4 > $O/c4_plain.json 2> $O/c4_plain.err; echo "plain rc=$?"
CNMFE_BENCH_FORCE_COLLECTIVES=1 timeout 300 python bench.py $X --config c4 > $O/c4_forced.json 2> $O/c4_forced.err; echo "forced rc=$?"
python - <<'PY'
import json
for f in ("c4_plain", "c4_forced"):
    try:
        j = json.loads(open("gpurun_out/r4c21/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms/step %.2f" % j["ms_per_step"], "kernel sum", j.get("kernel_sum_ms_per_step"), j.get("rccl_ranks"), j.get("backend"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/c4_forced.err
