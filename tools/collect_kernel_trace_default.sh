#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT bench command's headline leg (five contexts, SURVEY 8d's span, the vertex-block solver Model picks),
# condensed by tools/ktrace_stats.py.  Kernels of five contexts share the chip here: durations are those of the product's operating mode,
# not of a kernel alone (for that: tools/collect_kernel_trace.sh, one context).  An un-profiled run first fills the on-disk mesh cache.
# usage (GPU box, repo root): bash tools/collect_kernel_trace_default.sh OUT_PREFIX [bench args...]
set -e
OUT=$1; shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktd
ARGS="--no-extras --steps 1 --warmup 1 $@"
python3 $REPO/bench.py $ARGS > /tmp/bench_plain.json 2> /tmp/plain.err || { tail -5 /tmp/plain.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktd -- python3 $REPO/bench.py $ARGS > ${REPO}/${OUT}_bench_under_rocprof.json 2> /tmp/ktd.err || { tail -5 /tmp/ktd.err; exit 1; }
python3 $REPO/tools/ktrace_stats.py /tmp/ktd ${REPO}/${OUT}_kernel_stats_working.csv > ${REPO}/${OUT}_kernel_stats_working.txt; head -30 ${REPO}/${OUT}_kernel_stats_working.txt
