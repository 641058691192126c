#!/usr/bin/env python3
"""Timeline of the last complete `test` batch in a rocprofv3 --kernel-trace CSV (one line per launch):
   python tools/batch_trace.py gpurun_out/prof_x/t_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


starts = [i for i, r in enumerate(rows) if "k_sample_totals" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f  %-44s grid %sx%s" % ((s - t0) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"])[:44],
                                               r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", "")))
