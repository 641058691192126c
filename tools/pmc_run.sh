#!/bin/bash
# usage: tools/pmc_run.sh <outdir-tag> "<COUNTERS...>" <program args...>   (run via gpurun)
TAG=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $R/gpurun_out/$TAG -o pmc -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/$TAG.log 2>&1
cd $R
python3 - <<PY
import csv
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(int); dur=defaultdict(float)
for r in csv.DictReader(open("gpurun_out/$TAG/pmc_counter_collection.csv")):
    k=r["Kernel_Name"][:48]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    dur[(k,r["Dispatch_Id"])]=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
tot=defaultdict(float); cnt=defaultdict(int)
for (k,d),v in dur.items(): tot[k]+=v; cnt[k]+=1
for k in sorted(acc, key=lambda k:-tot[k])[:4]:
    print(k, "launches", cnt[k], "avg_ms %.3f"%(tot[k]/cnt[k]/1e6), {c: "%.4g"%(v/cnt[k]) for c,v in acc[k].items()})
PY
