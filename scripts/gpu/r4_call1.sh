#!/bin/bash
# round 4, GPU call 1: the queued ISA patches (scripts/probes/) as variant libraries, A/B against the baseline on one lease
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c1; mkdir -p $O
V=$PWD/cnmf_e_amd/variants
export PYTHONUNBUFFERED=1
run_tests() {  # name lib opts
  CNMFE_LIB=$2 CNMFE_OPTS=$3 timeout 900 python -m pytest tests -m gpu -x -q > $O/test_$1.txt 2>&1; echo "tests $1 rc=$?" | tee -a $O/summary.txt; tail -3 $O/test_$1.txt >> $O/summary.txt
}
bench() {  # tag lib opts args...
  local tag=$1 lib=$2 opts=$3; shift 3
  CNMFE_LIB=$lib CNMFE_OPTS=$opts timeout 300 python bench.py --no-extras --no-cpu-baseline "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?" >> $O/summary.txt
}
run_tests all $V/libcnmfe_all.so solve_gfill=1
if ! grep -q "rc=0" <(grep "tests all" $O/summary.txt); then
  run_tests gw $V/libcnmfe_gw.so solve_gfill=1
  run_tests gw0 $V/libcnmfe_gw.so solve_gfill=0
  run_tests rest $V/libcnmfe_rest.so ""
fi
for rep in 1 2; do
bench c3_base_$rep $V/libcnmfe_base.so "" --steps 20 --warmup 3
bench c3_all_$rep $V/libcnmfe_all.so solve_gfill=1 --steps 20 --warmup 3
done
bench c3_gw0 $V/libcnmfe_gw.so solve_gfill=0 --steps 20 --warmup 3
bench c3_gw1 $V/libcnmfe_gw.so solve_gfill=1 --steps 20 --warmup 3
bench c3_rest $V/libcnmfe_rest.so "" --steps 20 --warmup 3
bench demo_base $V/libcnmfe_base.so "" --steps 10 --warmup 2 --deconv --bg-ssub 2 --alg hals_thresh
bench demo_all $V/libcnmfe_all.so solve_gfill=1 --steps 10 --warmup 2 --deconv --bg-ssub 2 --alg hals_thresh
bench c4_base $V/libcnmfe_base.so "" --steps 10 --warmup 2 --config c4
bench c4_all $V/libcnmfe_all.so solve_gfill=1 --steps 10 --warmup 2 --config c4
python - <<'PY' >> gpurun_out/r4c1/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c1/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j.get("kernels_ms_per_step", {})
        print(f.split("bench_")[1][:-5], "ms/step %.2f" % j["ms_per_step"], {a: round(b, 3) for a, b in sorted(k.items(), key=lambda x: -x[1])[:12]})
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
