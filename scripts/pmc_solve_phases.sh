#!/bin/bash
export TMPDIR=/tmp
for sp in 0 1 8 16 4 2; do
  rm -rf /tmp/pm_x
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pm_x -o x -- python scripts/bg_only.py --cfg c3 --mode 3 --kernels 4 --sprobes $sp > /dev/null 2> /tmp/pm_x.err
  python - $sp <<'PY'
import csv,collections,glob,sys
agg=collections.defaultdict(list)
for f in glob.glob("/tmp/pm_x/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ring_solve2" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
d=[]
for f in glob.glob("/tmp/pm_x/**/*kernel_trace.csv", recursive=True):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(f)) if "k_ring_solve2" in r["Kernel_Name"]]
print("probe %s: ms %s  %s" % (sys.argv[1], [round(x,2) for x in d], {k: "%.3g"%(sum(v)/len(v)) for k,v in agg.items()}))
PY
done
