#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/tests20.txt 2>&1
tail -n 15 gpurun_out/r5/tests20.txt
