#!/bin/bash
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/gp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o g -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS > /tmp/gp.out 2>/tmp/gp.err
f=$(find /tmp/gp -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/scripts/gap_analysis.py $f ${GAP_MODE:-seq} ${GAP_NPI:-1} | tee $GRAFT_REPO_ROOT/gpurun_out/r03/gap_analysis_${GAP_TAG:-c3}.txt
