#!/bin/bash
# round 5, GPU call 12: kernel trace of the iteration -> gaps
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/r5; cd /tmp; rm -rf /tmp/gp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o g -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /tmp/gp.out 2>/tmp/gp.err
f=$(find /tmp/gp -name '*kernel_trace.csv' | head -1)
python $R/scripts/gap_analysis.py $f seq 1 > $R/gpurun_out/r5/gap_analysis_c3_v1.txt
python $R/scripts/gap_analysis.py $f agg 1 > $R/gpurun_out/r5/gap_analysis_c3_v1_agg.txt
python $R/scripts/host_timeline.py > $R/gpurun_out/r5/host_timeline_c3_v1.txt 2>&1
head -20 $R/gpurun_out/r5/gap_analysis_c3_v1.txt
