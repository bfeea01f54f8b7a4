#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c33; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
timeout 300 python scripts/host_timeline.py --patch 128 --iters 10 --force-collectives > $O/a.txt 2> $O/a.err; echo "forced timeline:"; grep -E "^iteration" $O/a.txt | awk '{printf "%s ", $3}'; echo
X="--no-extras --no-cpu-baseline --config c4 --steps 10 --warmup 4"
CNMFE_BENCH_FORCE_COLLECTIVES=1 timeout 300 python bench.py $X > $O/c4_forced.json 2> $O/c4_forced.err
timeout 300 python bench.py $X > $O/c4_plain.json 2> $O/c4_plain.err
CNMFE_BENCH_FORCE_COLLECTIVES=1 timeout 300 python bench.py $X --deconv > $O/c4_forced_deconv.json 2> $O/c4_forced_deconv.err
timeout 300 python bench.py $X --deconv > $O/c4_plain_deconv.json 2> $O/c4_plain_deconv.err
python - <<'PY'
import json
for f in ("c4_plain", "c4_forced", "c4_plain_deconv", "c4_forced_deconv"):
    try:
        j = json.loads(open("gpurun_out/r4c33/%s.json" % f).read().strip().splitlines()[-1]); print(f, "ms/step %.2f" % j["ms_per_step"], "kernel sum", j.get("kernel_sum_ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
