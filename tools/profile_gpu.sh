#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace statistics of bench.py and,
# with PMC=1, separate counter passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
# usage: [PMC=1] tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
echo "kernel-trace rc=$?"
grep '"metric"' "$OUT/bench.log" | cut -c1-400
if [ "${PMC:-0}" = "1" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o pmc -- python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/pmc_$C.log" 2>&1
    echo "pmc $C rc=$?"
  done
fi
find "$OUT" -name '*.csv' | head -20
