#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_summary.py DIR [substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_spmm"
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    for (kn, cn), (c, v) in sorted(agg.items()):
        print(f"{kn:42s} {cn:34s} n={c:4d} avg={v / c:.6g}")
