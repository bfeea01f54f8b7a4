#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c22; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python scripts/host_timeline.py --patch 128 --iters 4 --force-collectives --cprofile > $O/ht_forced.txt 2> $O/ht_forced.err; echo rc=$?
head -120 $O/ht_forced.txt | cut -c1-180
tail -3 $O/ht_forced.err
