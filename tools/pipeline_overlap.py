#!/usr/bin/env python3
"""Evidence for distributed.TestPipeline out of a rocprofv3 --kernel-trace of tools/gpu_pipeline_run.py:
   python3 tools/pipeline_overlap.py <trace.csv> <depth> <batches> <out.json>
the traced batches' wall span (first kernel start to last kernel end), the sum of their kernel durations, and how
much of the span had two or more kernels running."""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
depth, batches = int(sys.argv[2]), int(sys.argv[3])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the timed run is the last `batches` batches: count k_sample_totals launches from the end
starts = [i for i, e in enumerate(ev) if "k_sample_totals" in e[2]]
first = starts[-batches]
ev = ev[first:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = sum(e[1] - e[0] for e in ev)
pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
running, last, over1, over0 = 0, t0, 0, 0
for t, d in pts:
    if running >= 2:
        over1 += t - last
    if running >= 1:
        over0 += t - last
    running += d
    last = t
out = {"what": "rocprofv3 --kernel-trace of %d test batches (128 samples x 250 kb) through TestPipeline at depth %d" % (batches, depth),
       "depth": depth, "batches": batches, "span_ms": (t1 - t0) / 1e6, "ms_per_batch": (t1 - t0) / 1e6 / batches,
       "kernel_time_sum_ms": busy / 1e6, "kernel_time_over_span": busy / (t1 - t0),
       "span_fraction_with_a_kernel_running": over0 / (t1 - t0), "span_fraction_with_two_or_more_kernels_running": over1 / (t1 - t0),
       "kernels": len(ev)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
