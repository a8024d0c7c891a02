#!/usr/bin/env python3
"""Condense rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles-sized JSON.
usage: pmc_traffic.py FETCH_DIR WRITE_DIR BENCH_JSON OUT_JSON [RDREQ_DIR]   (run on the GPU box; the raw CSVs are large)
RDREQ_DIR: a pass with TCC_EA0_RDREQ_sum / _32B_sum / _128B_sum, which sizes the read requests: on gfx950 FETCH_SIZE
tallies every request at 64 B (MI355X_MICROARCH.md, HBM), so the read bytes are 32 n32 + 64 n64 + 128 n128.
The operator application of the CG = the kernels bench.py's event bracket encloses: k_spmm_pair (CSR), or k_patch_apply +
k_patch_dot (patch operator inside the PCG; the shared rows are summed by the update launch) - their traffic is added."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "remo::" in r["Kernel_Name"]:
                k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
    return {k: dict(launches=c, avg=v / c) for k, (c, v) in agg.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")     # KiB
sized = None
if len(sys.argv) > 5:
    tot, n32, n128 = (per_kernel(sys.argv[5], c) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum"))
    sized = {}
    for k in tot:
        t, a, b = tot[k]["avg"], n32.get(k, {"avg": 0})["avg"], n128.get(k, {"avg": 0})["avg"]
        sized[k] = dict(requests=t, req_32B=a, req_128B=b, read_bytes=32.0 * a + 128.0 * b + 64.0 * (t - a - b))
bench = json.load(open(sys.argv[3]))
cfg = bench["config"]
op = cfg.get("operator", "csr")
if op == "patch":
    keys = [k for k in fetch if k.startswith("remo::k_patch_apply<double")] + [k for k in fetch if k.startswith("remo::k_patch_dot")]
else:
    keys = [k for k in fetch if "k_spmm" in k and ", true" in k][:1]
parts = {}
for k in keys:
    rd = sized[k]["read_bytes"] if sized and k in sized else fetch[k]["avg"] * 1024.0
    parts[k] = dict(launches=fetch[k]["launches"], read_bytes=rd, write_bytes=write.get(k, {"avg": 0.0})["avg"] * 1024.0)
total = sum(p["read_bytes"] + p["write_bytes"] for p in parts.values())
size = re.search(r"mesh size (\w+)", cfg["workload"])
rf = bench.get("roofline", {})
out = dict(command="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_{sum,32B_sum,128B_sum} --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --depths 100 --batches 8 --streams 1 --resident --no-cpu --no-extras (three passes)",
           workload=cfg["workload"], mesh_size=size.group(1) if size else None, operator=op, mesh_T=cfg.get("mesh_T"), n_free=cfg["n_free"], nnz=cfg["nnz"],
           algorithmic_bytes_per_launch_same_launches=rf.get("algorithmic_bytes_per_launch"), launches_timed=rf.get("launches"),
           kernels=parts, spmm=dict(traffic_bytes_per_launch=total, kernels=keys),
           correction=("read bytes = 32 n32 + 64 n64 + 128 n128 from the TCC_EA0_RDREQ size counters (FETCH_SIZE = 64 B x requests under-counts the 128-B "
                       "requests of coalesced streams, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported") if sized else
                      "raw FETCH_SIZE + WRITE_SIZE (no request-size pass given)",
           all_kernels=dict(fetch_KiB={k: v["avg"] for k, v in fetch.items()}, write_KiB={k: v["avg"] for k, v in write.items()}, read_requests=sized))
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["spmm"]), json.dumps(parts))
