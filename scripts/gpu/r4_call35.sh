#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c35; mkdir -p $O
for rep in 1 2 3; do echo "process $rep"; python scripts/probes/malloc_probe/malloc_probe.py 1 4 10.5 10.5 11.5 3.8; done 2>&1 | tee $O/malloc_probe.txt
