#!/bin/bash
# round 5, GPU call 7: are the two video passes bound by their fp64 MFMAs?  (variant build without them); the gram_i8 edge tests
set -x
mkdir -p gpurun_out/r5; cd /root/repo
CNMFE_LIB=$PWD/cnmf_e_amd/variants/libcnmfe_nomfma.so timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench7_nomfma.json 2> gpurun_out/r5/bench7_nomfma.err
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5/bench7_ref.json 2> gpurun_out/r5/bench7_ref.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --demo-sequence > gpurun_out/r5/bench7_demo_seq.json 2> gpurun_out/r5/bench7_demo_seq.err
timeout 1200 python -m pytest tests/test_gpu_edges.py -x -q > gpurun_out/r5/tests7.txt 2>&1
tail -n 3 gpurun_out/r5/tests7.txt
