#!/bin/bash
# GPU box: counter passes (one --pmc set per run, kernel-trace only) over a short slice of the C4 trajectory.
# usage: scripts/r02_pmc.sh <tag>      -> gpurun_out/<tag>/pmc_<set>/...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_pmc}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc_$i" --output-format csv -- python3 "$ROOT/scripts/pmc_c4.py" 6 > "$OUT/pmc_$i.log" 2>&1
  echo "set $i ($set): rc $?" >> "$OUT/summary.txt"
done
find "$OUT" -name "*counter_collection.csv" | head -20 >> "$OUT/summary.txt"
