"""Timeline of the LAST repetition in a rocprofv3 kernel trace: per kernel start offset, duration, gap."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_sample_totals"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
lo = starts[-1]
seq = rows[lo:]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
tot_k = 0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")
    print("%8.2f us  dur %7.2f  gap %6.2f  grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g, r["Kernel_Name"][:70]))
    prev_end = max(prev_end, e)
    tot_k += e - s
print("kernels", len(seq), "span %.1f us" % ((prev_end - t0) / 1e3), "sum of kernel time %.1f us" % (tot_k / 1e3))
