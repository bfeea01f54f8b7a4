#!/bin/bash
mkdir -p gpurun_out
(
python scripts/deconv_phases.py --traces 100,300 2>&1 | grep -v "amdgpu\|^DT "
timeout 1500 python -m pytest tests -x -q -m gpu -k "deconv" 2>&1 | tail -5
) > gpurun_out/r6_call7.log 2>&1
tail -40 gpurun_out/r6_call7.log
