#!/bin/bash
# Run on the GPU box (through gpurun): the part of tools/refresh_profiles.sh that the `test` path and the bench line
# depend on (when only testpath.hip changed since the last full refresh): bench lines, the default bench's kernel
# statistics, the batched-test kernel statistics and batch times, the latency trace, the test path's counters.
#   gpurun --timeout 2400 -- 'bash tools/refresh_test_profiles.sh r06'
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$REPO"
PMC=0 bash tools/profile_gpu.sh ${TAG}_cfg2 --no-extra > "$OUT/cfg2_profile.log" 2>&1
P=$REPO/gpurun_out/prof_${TAG}_cfg2
cp "$(find $P/trace -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_cfg2_kernel_stats.csv"
grep '"metric"' $P/bench.log | tail -1 > "$OUT/${TAG}_cfg2_bench_under_rocprof.json"
timeout 1200 python3 bench.py 2> "$OUT/bench_stderr.log" | tail -1 > "$OUT/${TAG}_bench_cfg2_unprofiled.json"
cp bench_detail.json "$OUT/${TAG}_bench_detail.json" 2>/dev/null
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_cfg5" -o t -- python3 "$REPO/tools/gpu_test_scale.py" 125 50000 10 > "$OUT/cfg5_run.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_test250" -o t -- python3 "$REPO/tools/gpu_test_scale.py" 128 250000 20 > "$OUT/test250_run.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_cfg5_1000" -o t -- python3 "$REPO/tools/gpu_test_scale.py" 1000 50000 6 > "$OUT/cfg5_1000_run.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_test250_1000" -o t -- python3 "$REPO/tools/gpu_test_scale.py" 1000 250000 10 > "$OUT/test250_1000_run.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_lat" -o t -- python3 "$REPO/tools/gpu_lat_trace.py" 40 > "$OUT/lat_run.log" 2>&1 )
cp "$(find gpurun_out/prof_${TAG}_cfg5_1000 -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_cfg5_1000_samples_kernel_stats.csv"
cp "$(find gpurun_out/prof_${TAG}_test250_1000 -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_test250_1000_samples_kernel_stats.csv"
cp "$(find gpurun_out/prof_${TAG}_cfg5 -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_cfg5_test_kernel_stats.csv"
cp "$(find gpurun_out/prof_${TAG}_test250 -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_test250_kernel_stats.csv"
grep "batch of" "$OUT/cfg5_run.log" "$OUT/test250_run.log" "$OUT/cfg5_1000_run.log" "$OUT/test250_1000_run.log" > "$OUT/${TAG}_test_batch_times_under_rocprof.txt"
python3 tools/gpu_batch_trace.py "$(find gpurun_out/prof_${TAG}_cfg5 -name '*kernel_trace.csv' | head -1)" > "$OUT/${TAG}_batch_trace_125x50kb.txt" 2>&1
python3 tools/gpu_batch_trace.py "$(find gpurun_out/prof_${TAG}_test250 -name '*kernel_trace.csv' | head -1)" > "$OUT/${TAG}_batch_trace_128x250kb.txt" 2>&1
python3 - "$(find gpurun_out/prof_${TAG}_lat -name '*kernel_trace.csv' | head -1)" "$OUT/${TAG}_latency_trace.json" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
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
PY
bash tools/pmc_test_path.sh ${TAG} > "$OUT/pmc_test.log" 2>&1
cp profiles/${TAG}_test_pmc_busy.md profiles/${TAG}_test_pmc_busy.json "$OUT/" 2>/dev/null
ls "$OUT"
