"""Per-kernel totals and timeline of the LAST big test batch in a rocprofv3 kernel trace csv.
usage: python tools/batch_timeline.py <trace_kernel_trace.csv> [batch index from the end, default 1]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
idx = [i for i, r in enumerate(rows) if 'k_sample_totals' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
i0 = idx[-which - 1]
i1 = idx[-which]
t0 = int(rows[i0]['Start_Timestamp'])
tot = collections.OrderedDict()
for r in rows[i0:i1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    tot.setdefault(n, [0, 0.0])
    tot[n][0] += 1
    tot[n][1] += dur(r)
print("interval to next batch %.1f us" % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e3))
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %3d %8.1f us" % (n, c, d))
last = t0
gaps = 0.0
for r in rows[i0:i1]:
    g = (int(r['Start_Timestamp']) - last) / 1e3
    if g > 0:
        gaps += g
    last = max(last, int(r['End_Timestamp']))
print("idle gaps inside the batch: %.1f us" % gaps)
