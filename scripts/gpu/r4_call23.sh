#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c23; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
X="--no-extras --no-cpu-baseline --steps 10 --warmup 4"
timeout 300 python bench.py $X --config c4 > $O/c4_plain.json 2> $O/c4_plain.err
CNMFE_BENCH_FORCE_COLLECTIVES=1 timeout 300 python bench.py $X --config c4 > $O/c4_forced.json 2> $O/c4_forced.err
timeout 300 python scripts/host_timeline.py --patch 128 --iters 6 > $O/ht_plain.txt 2>&1
timeout 300 python scripts/host_timeline.py --patch 128 --iters 6 --force-collectives > $O/ht_forced.txt 2>&1
python - <<'PY'
import json
for f in ("c4_plain", "c4_forced"):
    try:
        j = json.loads(open("gpurun_out/r4c23/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms/step %.2f" % j["ms_per_step"], "kernel sum", j.get("kernel_sum_ms_per_step"), j.get("first_iteration", {}).get("warmup_steps_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -E "^iteration|last iteration" $O/ht_plain.txt $O/ht_forced.txt
