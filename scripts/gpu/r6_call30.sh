#!/bin/bash
# round 6, call 30: kernel timeline of one rank's share of c4 (two patches) -- where the 2 ms between kernel sum and iteration time go
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/gp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o g -- python $GRAFT_REPO_ROOT/scripts/rank_load.py --steps 4 > /tmp/gp.out 2>/tmp/gp.err
f=$(find /tmp/gp -name '*kernel_trace.csv' | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06
python $GRAFT_REPO_ROOT/scripts/gap_analysis.py $f seq 2 > $GRAFT_REPO_ROOT/gpurun_out/r06/gap_rank_load.txt
head -22 $GRAFT_REPO_ROOT/gpurun_out/r06/gap_rank_load.txt; tail -3 /tmp/gp.out
