#!/bin/bash
# usage (on the GPU box): tools/kstats.sh <tag> <python script> [args...]   -> per-kernel stats of the run, top 14 rows
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o t -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/$TAG/run.log 2>&1
cd $R
F=$(find gpurun_out/$TAG -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %5s avg_us %9.2f total_ms %8.3f %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
