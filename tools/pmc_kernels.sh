#!/bin/bash
# SQ counters of every kernel whose name contains one of the given substrings, one rocprofv3 --pmc pass over the first 4 batches of the
# headline sweep (their meshes must be in the on-disk cache: run tools/collect_kernel_trace.sh or the un-profiled command first).
# usage (GPU box, repo root): bash tools/pmc_kernels.sh "SUBSTR1,SUBSTR2" "COUNTER ..." [bench args...]
set -e
KERNELS=$1; COUNTERS=$2; shift 2
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout -k 10 300 rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d /tmp/pk -- python3 $REPO/bench.py --no-cpu --no-extras --steps 1 --warmup 0 --depths 100 --batches 4 --streams 1 --resident "$@" > /tmp/pk.json 2> /tmp/pk.err || { tail -5 /tmp/pk.err; exit 1; }
python3 - "$KERNELS" <<'PY'
import csv, glob, sys, collections
subs = sys.argv[1].split(",")
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pk/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if any(s in name for s in subs):
            a = agg[(name.split("(")[0][:60], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, cn), (c, v) in sorted(agg.items()):
    print("%-62s %-22s avg per launch %16.1f  (%d launches)" % (k, cn, v / c, c))
PY
