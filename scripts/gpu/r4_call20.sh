#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c20; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt
timeout 900 python bench.py > $O/bench_c3_v3.json 2> $O/bench_c3_v3.err; echo "bench rc=$?"
CNMFE_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err; echo "2 ranks rc=$?"
python - <<'PY'
import json
for f in ("bench_c3_v3", "bench_2ranks_one_device"):
    try:
        j = json.loads(open("gpurun_out/r4c20/%s.json" % f).read().strip().splitlines()[-1])
        print(f, j["value"], j["unit"], "ms/step", j["ms_per_step"], "n_gpus", j["n_gpus"], (j.get("roofline") or {}).get("frac"), (j.get("roofline") or {}).get("traffic"), j.get("rccl_ranks"))
    except Exception as e:
        print(f, "ERR", e)
PY
