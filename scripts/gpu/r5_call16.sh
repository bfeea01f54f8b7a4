#!/bin/bash
set -x
mkdir -p gpurun_out/r5; cd /root/repo
timeout 300 python scripts/solve_ab.py --cfg small --modes 0,1 --reps 2 > gpurun_out/r5/pair_small.txt 2>&1
timeout 400 python scripts/solve_ab.py --cfg c3 --modes 0,1 --reps 3 --probes 0,8 > gpurun_out/r5/pair_c3.txt 2>&1
cat gpurun_out/r5/pair_small.txt gpurun_out/r5/pair_c3.txt | grep -v "^+"
