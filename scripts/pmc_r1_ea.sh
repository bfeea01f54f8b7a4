#!/bin/bash
# memory-side (EA) request counters of one R1 sweep: what really leaves the L2 towards HBM
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for pass in "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_READ_sum TCC_WRITE_sum"; do
  rm -rf /tmp/pm_y
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_y -o y -- python $R/scripts/r1_probe.py --variant 11 --probes 0 --reps 1 > /tmp/pm_y.out 2> /tmp/pm_y.err || tail -n 2 /tmp/pm_y.err
  python - <<'PY'
import csv,collections,glob
for f in glob.glob("/tmp/pm_y/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_residual" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("%-28s last launch %.6g  (n=%d)"%(k,v[-1],len(v)))
PY
done
