#!/usr/bin/env python3
"""Condense rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles-sized JSON.
usage: pmc_traffic.py FETCH_DIR WRITE_DIR BENCH_JSON OUT_JSON   (run on the GPU box; the raw CSVs are large)"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "remo::" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
    return {k: dict(launches=c, avg_KiB=v / c) for k, (c, v) in agg.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
bench = json.load(open(sys.argv[3]))
cfg = bench["config"]
key = [k for k in fetch if "k_spmm" in k and ", true>" in k][0]
out = dict(command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu (two passes)",
           workload=cfg["workload"], n_free=cfg["n_free"], nnz=cfg["nnz"], kernel=key,
           spmm=dict(fetch_KiB=fetch[key]["avg_KiB"], write_KiB=write[key]["avg_KiB"], launches=fetch[key]["launches"],
                     traffic_bytes_per_launch=(fetch[key]["avg_KiB"] + write[key]["avg_KiB"]) * 1024.0),
           correction="none for the SpMM (TCC miss counts x 64 B match raw FETCH_SIZE for its 4-/8-byte per-lane loads, profiles/r01_b_pmc_spmm_sizeS.json); "
                      "the streaming vector kernels read 0.52x low and would need the guide's x2",
           fetch_KiB=fetch, write_KiB=write)
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["spmm"]))
