#!/bin/bash
# HBM traffic of the operator application of the bench workload: FETCH_SIZE / WRITE_SIZE / read-request-size passes (one counter
# group per pass, kernel trace only - never with the trace domains gpurun refuses), condensed on the box by tools/pmc_traffic.py.
# The profiled runs take the first 8 batches of the HEADLINE sweep itself (100 depths, mesh size L), one context, batches resident:
# an un-profiled run of the same command first leaves their meshes in the on-disk cache (meshgen.cached_mesh), so the profiled
# processes build nothing and start no child process behind the profiler's preloaded library.
# usage (GPU box, repo root): bash tools/collect_traffic.sh OUT_JSON [bench args...]
set -e
OUT=$(realpath -m $1); shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
export REMO_BENCH_TRACE_MESH=1
ARGS="--steps 1 --warmup 0 --depths 100 --batches 8 --streams 1 --resident --no-cpu --no-extras $@"
python3 $REPO/bench.py $ARGS > /tmp/bench_plain.json 2> /tmp/plain.err || { tail -5 /tmp/plain.err; exit 1; }
echo "un-profiled pass done (meshes cached)"
rm -rf /tmp/pf /tmp/pw /tmp/pr
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $REPO/bench.py $ARGS > /tmp/bench_pf.json 2> /tmp/pf.err || { tail -5 /tmp/pf.err; exit 1; }
echo "fetch pass done"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $REPO/bench.py $ARGS > /tmp/bench_pw.json 2> /tmp/pw.err || { tail -5 /tmp/pw.err; exit 1; }
echo "write pass done"
timeout -k 10 500 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d /tmp/pr -- python3 $REPO/bench.py $ARGS > /tmp/bench_pr.json 2> /tmp/pr.err || { tail -5 /tmp/pr.err; exit 1; }
echo "request-size pass done"
python3 $REPO/tools/pmc_traffic.py /tmp/pf /tmp/pw /tmp/bench_pf.json $OUT /tmp/pr
