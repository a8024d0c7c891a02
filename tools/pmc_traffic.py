#!/usr/bin/env python3
"""Condense rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles-sized JSON.
usage: pmc_traffic.py FETCH_DIR WRITE_DIR BENCH_JSON OUT_JSON [RDREQ_DIR]   (run on the GPU box; the raw CSVs are large)
RDREQ_DIR: a pass with TCC_EA0_RDREQ_sum / _32B_sum / _128B_sum, which sizes the read requests: on gfx950 FETCH_SIZE
tallies every request at 64 B (MI355X_MICROARCH.md, HBM), so the read bytes are 32 n32 + 64 n64 + 128 n128."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter, scale=1.0):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "remo::" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
    return {k: dict(launches=c, avg_KiB=v / c) for k, (c, v) in agg.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
sized = None
if len(sys.argv) > 5:
    tot, n32, n128 = (per_kernel(sys.argv[5], c) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum"))
    sized = {}
    for k in tot:
        t, a, b = tot[k]["avg_KiB"], n32.get(k, {"avg_KiB": 0})["avg_KiB"], n128.get(k, {"avg_KiB": 0})["avg_KiB"]   # "avg_KiB" holds plain request counts here
        sized[k] = dict(requests=t, req_32B=a, req_128B=b, read_bytes=32.0 * a + 128.0 * b + 64.0 * (t - a - b))
bench = json.load(open(sys.argv[3]))
cfg = bench["config"]
key = [k for k in fetch if "k_spmm" in k and ", true" in k][0]
out = dict(command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu (two passes)",
           workload=cfg["workload"], n_free=cfg["n_free"], nnz=cfg["nnz"], kernel=key,
           spmm=dict(fetch_KiB=fetch[key]["avg_KiB"], write_KiB=write[key]["avg_KiB"], launches=fetch[key]["launches"],
                     traffic_bytes_per_launch=((sized[key]["read_bytes"] if sized else fetch[key]["avg_KiB"] * 1024.0) + write[key]["avg_KiB"] * 1024.0)),
           correction=("read bytes = 32 n32 + 64 n64 + 128 n128 from the TCC_EA0_RDREQ size counters (FETCH_SIZE = 64 B x requests under-counts the 128-B "
                       "requests of coalesced streams, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported") if sized else
                      "raw FETCH_SIZE + WRITE_SIZE (no request-size pass given)",
           fetch_KiB=fetch, write_KiB=write, read_requests=sized)
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["spmm"]))
