#!/bin/bash
# round 5, GPU call 4: what bounds the ring solve?  PMC passes on the variants (0 = round 4, 2 = fused + look-ahead, 6 = shared diagonal steps, 9 = 6 + MFMA rank-2)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -io "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_ACTIVE[A-Z_0-9]*\|SQ_LDS[A-Z_0-9]*\|SQ_THREAD_CYCLES[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/pmc_counter_names.txt
cmd="python $R/scripts/solve_ab.py --cfg c3 --modes 0,2,6,9 --probes 0 --reps 2"
: > $O/pmc_solve_variants.txt
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_INSTS_BRANCH" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pm_r5
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_r5 -o x -- $cmd > /tmp/pm_r5.out 2> /tmp/pm_r5.err
  grep "solve_variant" /tmp/pm_r5.out | head -4 >> $O/pmc_solve_variants.txt
  python - <<'PY' >> $O/pmc_solve_variants.txt
import csv, collections, glob, re
fs = glob.glob("/tmp/pm_r5/**/*counter_collection.csv", recursive=True)
if not fs: print("no counters:", open("/tmp/pm_r5.err").read()[-600:])
for f in fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_ring_solve6<(\d+), *(\d+), *(\d+)>", r["Kernel_Name"])
        if m: agg["NT%s VAR%s WG%s" % m.groups()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key in sorted(agg):
        print("## " + key + "   " + "  ".join("%s %.5g" % (c, sum(v) / len(v)) for c, v in agg[key].items()))
PY
done
cat $O/pmc_solve_variants.txt
