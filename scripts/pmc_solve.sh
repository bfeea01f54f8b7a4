#!/bin/bash
# PMC passes on the per-pixel solve kernel at the headline size: bash scripts/pmc_solve.sh
export TMPDIR=/tmp
PASSES=${PASSES:-"0 1 2 3"}; i=-1
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAIT_INST_BR_MSG" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_REQ_sum"; do
  i=$((i+1)); [[ " $PASSES " == *" $i "* ]] || continue
  rm -rf /tmp/pm_x
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_x -o x -- python scripts/bg_only.py --cfg c3 --mode 3 --kernels ${GK:-4} "$@" > /dev/null 2> /tmp/pm_x.err
  python - <<'PY'
import csv,collections,glob
fs = glob.glob("/tmp/pm_x/**/*counter_collection.csv", recursive=True)
if not fs: print("no counters:", open("/tmp/pm_x.err").read()[-600:])
for f in fs:
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_ring_solve2" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("%-30s %.6g (n=%d)"%(k,sum(v)/len(v),len(v)))
for f in glob.glob("/tmp/pm_x/**/*kernel_trace.csv", recursive=True):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(f)) if "k_ring_solve2" in r["Kernel_Name"]]
    print("kernel duration under this pass (ms):", d)
PY
done
