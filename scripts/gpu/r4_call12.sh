#!/bin/bash
# cold-start stall hunt: the demo sequence from a fresh process, repeated, with the runtime's pin-in-place path for pageable transfers on (default) and off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c12; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
for rep in 1 2 3 4 5 6; do
  timeout 200 python bench.py --no-extras --no-cpu-baseline --demo-sequence > $O/seq_default_$rep.json 2>/dev/null
  GPU_PINNED_MIN_XFER_SIZE=1000000 timeout 200 python bench.py --no-extras --no-cpu-baseline --demo-sequence > $O/seq_nopin_$rep.json 2>/dev/null
done
python - <<'PY' > gpurun_out/r4c12/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c12/seq_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); k = j["kernels_ms_total"]
        big = {a: b for a, b in k.items() if b > 100 and a != "bg_gram_f64"}
        print(f.split("/")[-1], "%.3f s" % j["value"], big)
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
