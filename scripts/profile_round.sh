#!/bin/bash
# Round profile: kernel-trace stats + FETCH_SIZE / WRITE_SIZE PMC passes of `bench.py --steps 4 --warmup 1 --no-cpu-baseline`.
# Usage (on the GPU box): bash scripts/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/*
set -u
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp                      # (rocprofv3 wants a writable scratch directory as its working directory)
cmd="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $cmd > $out/bench_under_rocprof.json 2> $out/kt.err
cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- $cmd > /dev/null 2> $out/pmc_$c.err
  python - "$out" "$c" <<'PY'
import csv, collections, glob, sys
out, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/pmc_{c}.csv", "w") as g:
    g.write("counter,kernel,launches,value_per_launch_KiB,GB_per_launch\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        m = sum(v) / len(v)
        g.write('%s,"%s",%d,%.6g,%.4g\n' % (c, k[:90], len(v), m, m * 1024 / 1e9))
PY
  rm -rf $out/pmc_$c
done
rm -rf $out/kt
