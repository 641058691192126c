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
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_gpu.sh, PMC=1), KB per launch; "
        "bench.py doubles FETCH_SIZE as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950")
for wl in ("cfg2", "cfg4"):
    kernels = (("k_gram", "::k_gram<"), ("k_pick", "::k_pick"), ("k_rescore", "::k_rescore<"),
               ("k_finish", "::k_finish<"))
    vals = {name: {} for name, _ in kernels}
    for c in ("fetch_size", "write_size"):
        best = {}
        for row in csv.reader(open("%s/%s_%s_pmc_%s.csv" % (out, tag, wl, c))):
            if not row or row[0] == "Kernel":
                continue
            for kern, pat in kernels:
                # several instantiations may appear (the float32 variant is timed too): keep the most launched
                if pat in row[0] and int(row[1]) > best.get(kern, (0, 0.0))[0]:
                    best[kern] = (int(row[1]), float(row[2]))
        for kern, (_, v) in best.items():
            vals[kern][c] = v
    # the float64 re-score stage is k_pick + k_rescore (the pair engine); k_finish only runs under
    # WC_FINISH_ENGINE=rows
    stage = {}
    for c in ("fetch_size", "write_size"):
        parts = [vals[kk].get(c) for kk in ("k_pick", "k_rescore")]
        if all(v is not None for v in parts):
            stage[c] = sum(parts)
        elif vals["k_finish"].get(c) is not None:
            stage[c] = vals["k_finish"][c]
    res[wl] = {"fetch_kb_per_launch": vals["k_gram"].get("fetch_size"),
               "write_kb_per_launch": vals["k_gram"].get("write_size"),
               "k_finish": {"fetch_kb_per_launch": stage.get("fetch_size"),
                            "write_kb_per_launch": stage.get("write_size"),
                            "kernels": {kk: {"fetch_kb_per_launch": vals[kk].get("fetch_size"),
                                             "write_kb_per_launch": vals[kk].get("write_size")}
                                        for kk in ("k_pick", "k_rescore")}},
               "note": note}
json.dump(res, open("%s/%s_traffic.json" % (out, tag), "w"), indent=1)
print(json.dumps(res))
PY
ls -la "$OUT"
