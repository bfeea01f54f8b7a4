#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
CNMFE_OPTS=host_trace=1 timeout 300 python scripts/host_timeline.py --patch 128 --as-rank-of 8 > gpurun_out/r5/host_timeline_rank8.txt 2>&1
tail -n 30 gpurun_out/r5/host_timeline_rank8.txt
