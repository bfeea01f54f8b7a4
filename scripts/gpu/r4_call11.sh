#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c11; mkdir -p $O
export PYTHONUNBUFFERED=1 CNMFE_BENCH_R1=0
V=$PWD/cnmf_e_amd/variants
for v in default noinline oldsn; do
  L=""; [ $v != default ] && L=$V/libcnmfe_$v.so
  CNMFE_LIB=$L timeout 300 python bench.py --no-extras --no-cpu-baseline --config c5shard --steps 3 --warmup 2 --deconv > $O/bench_c5shard_$v.json 2> $O/bench_c5shard_$v.err
done
python - <<'PY' > gpurun_out/r4c11/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c11/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); k = j["kernels_ms_per_step"]
        print(f.split("bench_")[1][:-5], "ms/step %.2f" % j["ms_per_step"], {a: round(b, 2) for a, b in list(k.items())[:4]}, j["kernel_calls_per_step"].get("deconv_temporal"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
