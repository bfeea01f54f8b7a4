#!/bin/bash
# the search-mask thread at the end of the spatial update (CNMFE_PREFETCH_EARLY=1, rounds 3-6) against under the temporal sweep call: the headline, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for e in 1 0; do
CNMFE_PREFETCH_EARLY=$e timeout 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c3 prefetch early=$e:', round(d['ms_per_step'],3), 'kernel sum', d.get('kernel_sum_ms_per_step'))"
done; done
