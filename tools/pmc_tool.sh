#!/bin/bash
# Hardware counters of the kernels whose name contains SUBSTR, for any python tool of this repo: one rocprofv3 --pmc pass per
# ';'-separated counter group (counters of one group must fit one pass).
# usage (GPU box, repo root): bash tools/pmc_tool.sh OUT.txt "SUBSTR1,SUBSTR2" "C1 C2;C3 C4" tools/script.py [args...]
OUT=$1; KERNELS=$2; GROUPS_=$3; SCRIPT=$4; shift 4
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
: > $REPO/$OUT
IFS=';' read -ra GR <<< "$GROUPS_"
for g in "${GR[@]}"; do
  rm -rf /tmp/pk
  timeout -k 10 400 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pk -- python3 $REPO/$SCRIPT "$@" > /tmp/pk.out 2> /tmp/pk.err || { echo "pass [$g] failed:" >> $REPO/$OUT; tail -3 /tmp/pk.err >> $REPO/$OUT; continue; }
  python3 - "$KERNELS" >> $REPO/$OUT <<'PY'
import csv, glob, sys, collections, re
subs = sys.argv[1].split(",")
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pk/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if any(s in name for s in subs):
            short = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "").replace("remo::", "").split("(")[0]
            a = agg[(short[:60], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, cn), (c, v) in sorted(agg.items()):
    print("%-50s %-28s avg per launch %18.1f  (%d launches)" % (k, cn, v / c, c))
PY
done
