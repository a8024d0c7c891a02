#!/bin/bash
# rocprofv3 kernel trace of the first 8 batches of the headline sweep (one context, resident), condensed by tools/ktrace_stats.py (early-exit launches left out).
# usage (GPU box, repo root): bash tools/collect_kernel_trace.sh OUT_PREFIX [bench args...]
set -e
OUT=$1; shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
ARGS="--steps 2 --warmup 1 --depths 100 --batches 8 --streams 1 --resident --no-cpu --no-extras $@"
python3 $REPO/bench.py $ARGS > /tmp/bench_plain.json 2> /tmp/plain.err || { tail -5 /tmp/plain.err; exit 1; }     # fills the mesh cache: the profiled process starts no children
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $REPO/bench.py $ARGS > ${REPO}/${OUT}_bench_under_rocprof.json 2> /tmp/kt.err || { tail -5 /tmp/kt.err; exit 1; }
python3 $REPO/tools/ktrace_stats.py /tmp/kt ${REPO}/${OUT}_kernel_stats_working.csv | tee ${REPO}/${OUT}_kernel_stats_working.txt
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) ${REPO}/${OUT}_kernel_stats_raw.csv
