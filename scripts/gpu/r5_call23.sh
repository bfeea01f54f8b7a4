#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
CNMFE_OPTS=host_trace=1 timeout 300 python scripts/host_timeline.py --bg-ssub 2 > gpurun_out/r5/host_timeline_ssub2_c.txt 2>&1
grep -v "launch " gpurun_out/r5/host_timeline_ssub2_c.txt | tail -n 75 | head -62
