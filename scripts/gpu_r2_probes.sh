#!/bin/bash
# Evidence files for DESIGN.md section 3 (round 2): R1 timing ablations + TCC counters, solve A/B, host timelines, single-rank nccl smoke.
o=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $o
cd $GRAFT_REPO_ROOT
python scripts/r1_probe.py --variant 11 --probes 0,1,2,3,7,15,16 > $o/r1_probe_v11.txt 2>/dev/null
python scripts/r1_probe.py --variant 12 --probes 0,1,2,3,16 > $o/r1_probe_v12.txt 2>/dev/null
bash scripts/pmc_r1_probe.sh "11 12" 0 > $o/r1_pmc_tcc.txt 2>&1
cd $GRAFT_REPO_ROOT
python scripts/solve_ab.py --cfg c3 --modes 2,5,6 --probes 0,1,2,4 > $o/solve_ab_c3.txt 2>/dev/null
python scripts/host_timeline.py --iters 4 > $o/host_timeline_c3.txt 2>/dev/null
python scripts/host_timeline.py --iters 4 --patch 128 > $o/host_timeline_c4.txt 2>/dev/null
python scripts/host_trace_fit.py > /dev/null 2> $o/host_trace_fit_c3.txt
python scripts/nccl_smoke.py > $o/nccl_smoke.txt 2>&1
tail -n 4 $o/r1_probe_v11.txt $o/r1_probe_v12.txt; grep -c variant $o/r1_pmc_tcc.txt; tail -n 6 $o/solve_ab_c3.txt; tail -n 3 $o/nccl_smoke.txt
