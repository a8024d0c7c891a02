#!/bin/bash
# rocprofv3 kernel trace of any python tool of this repo, condensed by tools/ktrace_stats.py.
# usage (GPU box, repo root): bash tools/trace_tool.sh OUT_PREFIX tools/script.py [args...]
set -e
OUT=$1; shift
SCRIPT=$1; shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $REPO/$SCRIPT "$@" > ${REPO}/${OUT}_tool_output.log 2> /tmp/kt.err || { tail -5 /tmp/kt.err; exit 1; }
python3 $REPO/tools/ktrace_stats.py /tmp/kt ${REPO}/${OUT}_kernel_stats_working.csv > ${REPO}/${OUT}_kernel_stats_working.txt
