#!/bin/bash
# host profile of one rank's iteration (2 of 16 patches): where the Python of a rank goes
cd $GRAFT_REPO_ROOT
timeout 280 python scripts/rank_load.py --world 8 --steps 10 > gpurun_out/rank_load_v4.txt 2>&1
timeout 280 python -c "
import cProfile, pstats, sys, runpy
sys.argv = ['rank_load.py', '--world', '8', '--steps', '30']
pr = cProfile.Profile(); pr.enable()
try:
    runpy.run_path('scripts/rank_load.py', run_name='__main__')
finally:
    pr.disable()
    st = pstats.Stats(pr, stream=open('gpurun_out/rank_prof_cum.txt', 'w')); st.sort_stats('cumulative').print_stats(70)
    st = pstats.Stats(pr, stream=open('gpurun_out/rank_prof_tot.txt', 'w')); st.sort_stats('tottime').print_stats(50)
" > gpurun_out/rank_prof.log 2>&1
tail -3 gpurun_out/rank_load_v4.txt | cut -c1-300
