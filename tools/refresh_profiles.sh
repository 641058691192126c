#!/bin/bash
# Run on the GPU box (through gpurun): everything the committed profiles/ files are made of.
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r01'
# Writes gpurun_out/profiles_<tag>/: kernel stats of the default bench (cfg2) and of
# --workload cfg4, FETCH_SIZE / WRITE_SIZE summaries (separate PMC passes), the bench lines
# seen under the profiler and one unprofiled default bench line, and the traffic json.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$REPO"
for WL in cfg2 cfg4; do
  ARGS="--no-extra"
  [ "$WL" = cfg4 ] && ARGS="--workload cfg4 --steps 4 --warmup 1 --no-extra"
  PMC=1 bash tools/profile_gpu.sh ${TAG}_$WL $ARGS > "$OUT/${WL}_profile.log" 2>&1
  P=$REPO/gpurun_out/prof_${TAG}_$WL
  cp "$(find $P/trace -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_${WL}_kernel_stats.csv"
  grep '"metric"' $P/bench.log | tail -1 > "$OUT/${TAG}_${WL}_bench_under_rocprof.json"
  for C in FETCH_SIZE WRITE_SIZE; do
    lc=$(echo $C | tr 'A-Z' 'a-z')
    python3 tools/pmc_summary.py "$(find $P/pmc_$C -name '*counter_collection.csv' | head -1)" > "$OUT/${TAG}_${WL}_pmc_${lc}.csv"
  done
done
python3 bench.py 2> "$OUT/bench_stderr.log" | tail -1 > "$OUT/${TAG}_bench_cfg2_unprofiled.json"
python3 - "$OUT" "$TAG" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for wl in ("cfg2", "cfg4"):
    vals = {}
    for c in ("fetch_size", "write_size"):
        for row in csv.reader(open("%s/%s_%s_pmc_%s.csv" % (out, tag, wl, c))):
            if row and "::k_gram" in row[0] and "thr16" not in row[0]:
                vals[c] = float(row[2])
    res[wl] = {"fetch_kb_per_launch": vals.get("fetch_size"), "write_kb_per_launch": vals.get("write_size"),
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_gpu.sh, PMC=1), "
                       "KB per k_gram launch; bench.py doubles FETCH_SIZE as MI355X_MICROARCH.md prescribes for "
                       "16 B/lane streaming reads on gfx950"}
json.dump(res, open("%s/%s_traffic.json" % (out, tag), "w"), indent=1)
print(json.dumps(res))
PY
ls -la "$OUT"
