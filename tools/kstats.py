#!/usr/bin/env python3
"""Short per-kernel table from a rocprofv3 --kernel-trace --stats run: python tools/kstats.py kernel_stats.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r["Name"]
    m = re.search(r"(remo::(?:\(anonymous namespace\)::)?\w+(?:<[^>(]*>)?)", name)
    short = m.group(1) if m else ("rocprim::" + (re.search(r"(radix_sort\w*|partition\w*|merge_sort\w*|scan\w*|transform\w*|unique\w*|reduce\w*)", name) or [None, "other"])[1] if "rocprim" in name else name[:48])
    print(f"{short:58s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  {float(r['Percentage']):5.2f} %")
