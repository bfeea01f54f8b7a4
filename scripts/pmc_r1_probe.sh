#!/bin/bash
# PMC passes on the R1 kernels under scripts/r1_probe.py: bash scripts/pmc_r1_probe.sh "<variants>" "<probe>"
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for v in $1; do
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf /tmp/pm_x
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_x -o x -- python $R/scripts/r1_probe.py --variant $v --probes $2 --reps 1 > /tmp/pm_x.out 2> /tmp/pm_x.err || tail -n 3 /tmp/pm_x.err
  python - "$v" <<'PY'
import csv,collections,glob,sys
for f in glob.glob("/tmp/pm_x/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_residual" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("variant %s  %-30s last launch %.6g  (n=%d)"%(sys.argv[1],k,v[-1],len(v)))
PY
done
done
