#!/bin/bash
# PMC passes on one R1 variant: bash scripts/pmc_r1.sh <variant> [--ac]
export TMPDIR=/tmp
v=$1; shift
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES" "FETCH_SIZE"; do
  rm -rf /tmp/pm_x
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_x -o x -- python scripts/r1_only.py --variant $v --reps 1 "$@" > /dev/null 2> /tmp/pm_x.err
  python - <<'PY'
import csv,collections,glob
for f in glob.glob("/tmp/pm_x/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_residual" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("%-28s %.6g (n=%d)"%(k,sum(v)/len(v),len(v)))
PY
done
