#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r5/tests30.txt 2>&1
tail -n 22 gpurun_out/r5/tests30.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
