#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c13; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
for rep in 1 2 3 4 5 6 7 8; do
  CNMFE_OPTS=host_trace=1 timeout 200 python bench.py --no-extras --no-cpu-baseline --demo-sequence > $O/seq_$rep.json 2> $O/seq_$rep.err
done
python - <<'PY' > gpurun_out/r4c13/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c13/seq_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.3f s" % j["value"], j["host_return_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
