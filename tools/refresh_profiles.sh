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
timeout 1200 python3 bench.py 2> "$OUT/bench_stderr.log" | tail -1 > "$OUT/${TAG}_bench_cfg2_unprofiled.json"
timeout 600 python3 bench.py --workload cfg4 --steps 10 --warmup 2 --no-extra --no-cpu-baseline 2> "$OUT/bench_cfg4_stderr.log" | tail -1 > "$OUT/${TAG}_bench_cfg4_unprofiled.json"
# batched `test` at 50 kb (config 5's per-GPU share) and one sample per call (config 3): kernel statistics
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_cfg5" -o t -- python3 "$REPO/tools/gpu_test_scale.py" 125 50000 10 > "$OUT/cfg5_run.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_test250" -o t -- python3 "$REPO/tools/gpu_test_scale.py" 128 250000 20 > "$OUT/test250_run.log" 2>&1
  # the north-star's calls: 1 000 samples in one wc_test_batch_dev call at both bin sizes
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_cfg5_1000" -o t -- python3 "$REPO/tools/gpu_test_scale.py" 1000 50000 6 > "$OUT/cfg5_1000_run.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_test250_1000" -o t -- python3 "$REPO/tools/gpu_test_scale.py" 1000 250000 10 > "$OUT/test250_1000_run.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_lat" -o t -- python3 "$REPO/tools/gpu_lat_trace.py" 40 > "$OUT/lat_run.log" 2>&1 )
cp "$(find gpurun_out/prof_${TAG}_cfg5_1000 -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_cfg5_1000_samples_kernel_stats.csv"
cp "$(find gpurun_out/prof_${TAG}_test250_1000 -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_test250_1000_samples_kernel_stats.csv"
grep "batch of" "$OUT/cfg5_run.log" "$OUT/test250_run.log" "$OUT/cfg5_1000_run.log" "$OUT/test250_1000_run.log" > "$OUT/${TAG}_test_batch_times_under_rocprof.txt"
# the gather roof of the z-score stage: micro-benchmark by matrix size and access width, and k_zscore's TA / TCP / TLB counters
( for n in 3000 11087 55337; do timeout 120 python3 tools/micro/gather_rate.py $n 2>&1 | grep "^bins" | head -5; done ) > "$OUT/${TAG}_gather_roof.txt"
( echo "== 125 x 50 kb"; PMC_TIMEOUT=120 bash tools/pmc_zscore.sh ${TAG} 125 50000; echo "== 128 x 250 kb"; PMC_TIMEOUT=120 bash tools/pmc_zscore.sh ${TAG}q 128 250000 ) > "$OUT/${TAG}_zscore_ta_tcp.txt" 2>&1
cp "$(find gpurun_out/prof_${TAG}_cfg5 -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_cfg5_test_kernel_stats.csv"
cp "$(find gpurun_out/prof_${TAG}_test250 -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_test250_kernel_stats.csv"
python3 - "$(find gpurun_out/prof_${TAG}_lat -name '*kernel_trace.csv' | head -1)" "$OUT/${TAG}_latency_trace.json" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
# one latency-mode call = the launches from k_lat_project to k_assemble_calls; average the last 20 calls
starts = [i for i, r in enumerate(rows) if "k_lat_project" in r["Kernel_Name"]]
calls = []
for a, b in zip(starts[-21:-1], starts[-20:]):
    seq = rows[a:b]
    calls.append(([(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in seq],
                  (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3))
names = [n for n, _ in calls[-1][0]]
same = [c for c in calls if [n for n, _ in c[0]] == names]
per = [{"kernel": names[i], "us": sum(c[0][i][1] for c in same) / len(same)} for i in range(len(names))]
json.dump({"what": "one latency-mode test call (one 250 kb sample) under rocprofv3 --kernel-trace: mean over %d replays of the "
                   "captured graph" % len(same),
           "launches": len(names), "kernels": per, "kernel_us_sum": sum(p["us"] for p in per),
           "span_us_first_start_to_last_end": sum(c[1] for c in same) / len(same)}, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
# newref prep at 600 x 50 kb (Gram, eigen-solve on the GPU, finish): kernel statistics and the eigen-solver's timings
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_prep" -o t -- python3 "$REPO/tools/gpu_prep_time.py" cfg4 > "$OUT/prep_run.log" 2>&1 )
cp "$(find gpurun_out/prof_${TAG}_prep -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_prep_cfg4_kernel_stats.csv"
python3 tools/gpu_prep_time.py cfg4 > "$OUT/${TAG}_prep_cfg4_times.txt"
python3 tools/gpu_eig_time.py 100 300 600 1000 1200 2400 > "$OUT/${TAG}_eig_times.json" 2> /dev/null
# busy / cache counters of the newref kernels (one pass per counter set)
bash tools/pmc_run.sh ${TAG}A_cfg2 "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/gpu_newref_only.py cfg2 10 > "$OUT/pmcA_cfg2.log" 2>&1
bash tools/pmc_run.sh ${TAG}A_cfg4 "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/gpu_newref_only.py cfg4 3 > "$OUT/pmcA_cfg4.log" 2>&1
bash tools/pmc_run.sh ${TAG}T_cfg2 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" tools/gpu_newref_only.py cfg2 10 > "$OUT/pmcT_cfg2.log" 2>&1
bash tools/pmc_run.sh ${TAG}T_cfg4 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" tools/gpu_newref_only.py cfg4 3 > "$OUT/pmcT_cfg4.log" 2>&1
bash tools/pmc_run.sh ${TAG}W_cfg2 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" tools/gpu_newref_only.py cfg2 10 > "$OUT/pmcW_cfg2.log" 2>&1
bash tools/pmc_run.sh ${TAG}W_cfg4 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" tools/gpu_newref_only.py cfg4 3 > "$OUT/pmcW_cfg4.log" 2>&1
# the same three counter sets for the `test` path kernels (128 x 250 kb and 125 x 50 kb batches)
bash tools/pmc_test_path.sh ${TAG} > "$OUT/pmc_test.log" 2>&1
python3 tools/busy_summary.py $TAG gpurun_out/${TAG}A_cfg2 gpurun_out/${TAG}A_cfg4 gpurun_out/${TAG}T_cfg2 gpurun_out/${TAG}T_cfg4 gpurun_out/${TAG}W_cfg2 gpurun_out/${TAG}W_cfg4 > "$OUT/busy.log" 2>&1
cp profiles/${TAG}_pmc_busy.md profiles/${TAG}_pmc_busy.json profiles/${TAG}_test_pmc_busy.md profiles/${TAG}_test_pmc_busy.json "$OUT/" 2>/dev/null
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
                # several instantiations may appear (the other tile modes are timed too): keep the most launched
                hit = pat in row[0] or (kern == "k_gram" and "::k_gram_glds" in row[0])
                if hit and int(row[1]) > best.get(kern, (0, 0.0))[0]:
                    best[kern] = (int(row[1]), float(row[2]))
        for kern, (_, v) in best.items():
            vals[kern][c] = v
    # the float64 re-score stage is k_pick + k_rescore (the pair engine); k_finish only runs beyond 2048 samples
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
