#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel (CSV output)."""
import csv
import sys
from collections import defaultdict


def main(path):
    acc = defaultdict(lambda: [0, 0.0])
    name = None
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        acc[k][0] += 1
        acc[k][1] += float(row["Counter_Value"])
        name = row["Counter_Name"]
    print("Kernel,Calls,Avg_%s" % name)
    for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%.1f' % (k, n, tot / n))


if __name__ == "__main__":
    main(sys.argv[1])
