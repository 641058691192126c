"""The launches of ONE lone batch in time order (from a rocprofv3 --kernel-trace csv of tools/gpu_test_ab.py):
    python tools/gpu_batch_trace.py <kernel_trace.csv>     -> start offset, duration, gap to the previous end, per stream"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
# the last batch: from the last k_sample_totals on
starts = [i for i, r in enumerate(rows) if "k_sample_totals" in r["Kernel_Name"]]
seq = rows[starts[-2]:starts[-1]]
cut = [i for i, r in enumerate(seq) if "at::native" in r["Kernel_Name"]]      # (the harness's own torch kernels behind the call)
if cut: seq = seq[:cut[0]]
t0 = int(seq[0]["Start_Timestamp"])
last_end = {}
print("batch span %.1f us, %d launches" % ((max(int(r["End_Timestamp"]) for r in seq) - t0) / 1e3, len(seq)))
for r in seq:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("q%-3s %8.1f us  +%6.1f  dur %7.1f  %s" % (q, (s - t0) / 1e3, gap, (e - s) / 1e3, short(r["Kernel_Name"])))
