#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace run with the EARLY-EXIT launches left out.

The host queues PCG launches a few steps ahead of the device; once every column of a solve is frozen the launches that are
already queued return at once (kernels.hip solve_done: 3-4 us instead of 50).  rocprofv3 --stats averages them in, which
understates the working kernel.  This script reads the per-dispatch trace (*_kernel_trace.csv), drops, kernel by kernel,
the dispatches shorter than `--floor` (default 0.35) of that kernel's MEDIAN duration, and prints / writes both views.

usage: ktrace_stats.py TRACE_DIR_OR_CSV [OUT.csv] [--floor 0.35]"""
import csv
import glob
import os
import re
import statistics
import sys


def short(name):
    m = re.search(r"(remo::(?:\(anonymous namespace\)::)?\w+(?:<[^>(]*>)?)", name)
    if m:
        return m.group(1)
    if "rocprim" in name:
        k = re.search(r"(radix_sort\w*|partition\w*|merge_sort\w*|scan\w*|transform\w*|unique\w*|reduce\w*|lookback\w*)", name)
        return "rocprim::" + (k.group(1) if k else "other")
    return name[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    floor = 0.35
    if "--floor" in sys.argv:
        floor = float(sys.argv[sys.argv.index("--floor") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--floor") + 1]]
    src = args[0]
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    durs = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            durs.setdefault(short(r["Kernel_Name"]), []).append(d)
    rows = []
    for k, v in durs.items():
        med = statistics.median(v)
        work = [d for d in v if d >= floor * med]
        rows.append(dict(kernel=k, calls=len(v), avg_us=sum(v) / len(v), working_calls=len(work), early_exit_calls=len(v) - len(work),
                         working_avg_us=sum(work) / len(work), working_min_us=min(work), working_max_us=max(work), median_us=med,
                         working_total_ms=sum(work) / 1e3))
    rows.sort(key=lambda r: -r["working_total_ms"])
    tot = sum(r["working_total_ms"] for r in rows)
    for r in rows:
        r["share"] = r["working_total_ms"] / tot
        print("%-62s calls %6d (early exits %5d)  all-avg %8.2f us  WORKING avg %8.2f us  min %7.2f  max %8.2f  total %8.2f ms  %5.1f %%" %
              (r["kernel"], r["calls"], r["early_exit_calls"], r["avg_us"], r["working_avg_us"], r["working_min_us"], r["working_max_us"],
               r["working_total_ms"], 100 * r["share"]))
    if len(args) > 1:
        with open(args[1], "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)


if __name__ == "__main__":
    main()
