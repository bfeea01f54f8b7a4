#!/bin/bash
set -x
mkdir -p gpurun_out/r5; cd /root/repo
CNMFE_LIB=$PWD/cnmf_e_amd/variants/libcnmfe_nomfma.so timeout 300 python scripts/video_pass_probe.py > gpurun_out/r5/video_pass_probe_nomfma.txt 2>&1
timeout 300 python scripts/video_pass_probe.py > gpurun_out/r5/video_pass_probe_ref.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extras --demo-sequence > gpurun_out/r5/bench8_demo_seq.json 2> gpurun_out/r5/bench8_demo_seq.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --demo-sequence > gpurun_out/r5/bench8_demo_seq_b.json 2> gpurun_out/r5/bench8_demo_seq_b.err
timeout 1200 python -m pytest tests/test_gpu_edges.py -x -q > gpurun_out/r5/tests8.txt 2>&1
tail -n 3 gpurun_out/r5/tests8.txt; grep "ms per call\|raised" gpurun_out/r5/video_pass_probe_*.txt
