#!/bin/bash
# PMC passes over the SpMM-only driver (one counter group per pass; no trace domains besides the kernel trace).
# usage (on the GPU box, from the repo root): bash tools/pmc_spmm_passes.sh VARIANT OUTFILE
set -e
V=$1
OUT=$2
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" \
  "TCC_BUSY_avr TCC_REQ_sum TCC_TAG_STALL_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  echo "pass $i: $grp"
  timeout -k 5 100 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$i -- python3 $REPO/tools/spmm_only.py S 5 $V 16 256 0 1024 30 > /tmp/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pmc_$i.log; continue; }
  python3 $REPO/tools/pmc_summary.py /tmp/pmc_$i k_spmm >> $OUT
  echo "pass $i done"
done
