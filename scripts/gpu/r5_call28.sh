#!/bin/bash
mkdir -p gpurun_out/r5; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r5/tests28.txt 2>&1
tail -n 60 gpurun_out/r5/tests28.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
