#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c10; mkdir -p $O
export PYTHONUNBUFFERED=1
( cd /tmp && export TMPDIR=/tmp && CNMFE_BENCH_R1=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace.err )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/gap_analysis.py "$f" seq > $O/gap_analysis_c3_defer.txt 2>&1
rm -rf $O/trace
CNMFE_OPTS=host_trace=1 CNMFE_BENCH_R1=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 2 > $O/bench_ht.json 2> $O/bench_ht.err
python scripts/host_timeline.py > $O/host_timeline.txt 2>&1
head -20 $O/gap_analysis_c3_defer.txt
