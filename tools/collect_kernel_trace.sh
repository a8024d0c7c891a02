#!/bin/bash
# rocprofv3 kernel trace of a short bench run, condensed by tools/ktrace_stats.py (early-exit launches left out).
# usage (GPU box, repo root): bash tools/collect_kernel_trace.sh OUT_PREFIX [bench args...]
set -e
OUT=$1; shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
REMO_BENCH_TRACE_MESH=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $REPO/bench.py --no-cpu --no-extras --mesh-workers 1 --depths 20 "$@" > ${REPO}/${OUT}_bench_under_rocprof.json 2> /tmp/kt.err || { tail -5 /tmp/kt.err; exit 1; }
python3 $REPO/tools/ktrace_stats.py /tmp/kt ${REPO}/${OUT}_kernel_stats_working.csv | tee ${REPO}/${OUT}_kernel_stats_working.txt
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) ${REPO}/${OUT}_kernel_stats_raw.csv
