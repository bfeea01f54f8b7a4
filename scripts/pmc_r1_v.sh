#!/bin/bash
# PMC passes (SQ, TCC, FETCH/WRITE) on one R1 variant at H: bash scripts/pmc_r1_v.sh <variant> [probe]
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
v=$1; pr=${2:-0}
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm_x
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_x -o x -- python $R/scripts/r1_probe.py --variant $v --probes $pr --reps 1 > /tmp/pm_x.out 2> /tmp/pm_x.err || tail -n 3 /tmp/pm_x.err
  python - "$v" <<'PY'
import csv,collections,glob,sys
for f in glob.glob("/tmp/pm_x/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_residual" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("variant %s  %-30s last launch %.6g  (n=%d)"%(sys.argv[1],k,v[-1],len(v)))
PY
done
