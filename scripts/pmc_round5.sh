#!/bin/bash
# PMC passes (pipe utilisation) on the iteration's three large kernels at the headline size: bash scripts/pmc_round5.sh  -> gpurun_out/pmc_r05/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_r05; mkdir -p $O
export TMPDIR=/tmp CNMFE_BENCH_R1=0; cd /tmp
cmd="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
: > $O/summary.txt
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pm_r5
  timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_r5 -o x -- $cmd > /dev/null 2> /tmp/pm_r5.err
  python - <<'PY' >> $O/summary.txt
import csv, collections, glob
names = {"k_ring_solve6": "bg_ring_solve (k_ring_solve6<6>)", "k_vp_proj_i8": "temporal_proj_B (k_vp_proj_i8<1>)", "k_vp_proj_b": "temporal_proj_B (k_vp_proj_b<1, true>, proj_i8 = 0)", "k_win_proj_i8": "bg_win_proj (k_win_proj_i8)"}
fs = glob.glob("/tmp/pm_r5/**/*counter_collection.csv", recursive=True)
if not fs: print("no counters:", open("/tmp/pm_r5.err").read()[-400:])
for f in fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for key in names:
            if key in r["Kernel_Name"]: agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key in names:
        if agg[key]:
            print("## " + names[key])
            for c, v in agg[key].items(): print("   %-28s %.6g  (mean of %d launches)" % (c, sum(v) / len(v), len(v)))
PY
done
cat $O/summary.txt
