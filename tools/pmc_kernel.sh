#!/bin/bash
# SQ counters of one kernel of a short bench run (one rocprofv3 --pmc pass, kernel trace only).
# usage (GPU box, repo root): bash tools/pmc_kernel.sh KERNEL_SUBSTRING "COUNTER ..." [bench args...]
set -e
KERNEL=$1; COUNTERS=$2; shift 2
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout -k 10 300 rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d /tmp/pk -- python3 $REPO/bench.py --no-cpu --no-extras --mesh-workers 1 --steps 1 --warmup 0 --depths 10 "$@" > /tmp/pk.json 2> /tmp/pk.err || { tail -5 /tmp/pk.err; exit 1; }
python3 - "$KERNEL" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pk/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (c, v) in sorted(agg.items()):
    print("%-28s avg per launch %14.1f  (%d launches)" % (k, v / c, c))
PY
