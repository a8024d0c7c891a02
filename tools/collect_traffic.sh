#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / read-request-size passes of the default bench workload (one counter per pass, kernel trace only),
# condensed on the box by tools/pmc_traffic.py.  usage (GPU box, repo root): bash tools/collect_traffic.sh OUT_JSON
set -e
OUT=$(realpath -m $1)
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
( while true; do sleep 45; echo "heartbeat $(date +%T)"; done ) &
HB=$!
trap "kill $HB" EXIT
rm -rf /tmp/pf /tmp/pw
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --mesh-workers 1 > /tmp/bench_pf.json 2> /tmp/pf.err
echo "fetch pass done"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --mesh-workers 1 > /tmp/bench_pw.json 2> /tmp/pw.err
echo "write pass done"
rm -rf /tmp/pr
timeout -k 10 500 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d /tmp/pr -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --mesh-workers 1 > /tmp/bench_pr.json 2> /tmp/pr.err
echo "request-size pass done"
python3 $REPO/tools/pmc_traffic.py /tmp/pf /tmp/pw /tmp/bench_pf.json $OUT /tmp/pr
