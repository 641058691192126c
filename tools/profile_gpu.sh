#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace statistics of bench.py, summaries into gpurun_out/.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o trace -- python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
echo "rocprofv3 rc=$?"
tail -1 "$OUT/bench.log" | cut -c1-600
find "$OUT" -name '*kernel_stats*' | head
F=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && head -25 "$F"
