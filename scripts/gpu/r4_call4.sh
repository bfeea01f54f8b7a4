#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c4; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_zconfigs.py -x -q -k "demo_defaults or c3_full" --durations=5 > $O/test_new.txt 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt; tail -40 $O/test_new.txt >> $O/summary.txt
cat $O/summary.txt
