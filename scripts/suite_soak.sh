#!/bin/bash
# GPU box: repeated runs of the GPU suite with faulthandler and core files enabled (VERDICT r3 item 2 / ADVICE r3: one unexplained abort in round 3).
#   usage: scripts/suite_soak.sh TAG FULL_RUNS XCD_RUNS   ->  gpurun_out/TAG/tally.txt (+ the full output of any run that did not end in "passed")
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-soak}; FULL=${2:-2}; XCD=${3:-6}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
ulimit -c unlimited
export PYTHONFAULTHANDLER=1
cd "$ROOT"
run() { # label, pytest args...
  local label=$1; shift
  local t0=$(date +%s)
  python -X faulthandler -m pytest "$@" -m gpu -q -p no:cacheprovider > "$OUT/$label.log" 2>&1
  local rc=$?
  local line=$(grep -a -E "passed|failed|error" "$OUT/$label.log" | tail -1)
  echo "$label rc=$rc $(( $(date +%s) - t0 )) s: $line" >> "$OUT/tally.txt"
  if [ $rc -eq 0 ]; then rm -f "$OUT/$label.log"; else ls core* 2>/dev/null >> "$OUT/tally.txt"; fi
}
for i in $(seq 1 $FULL); do run full_$i tests; done
for i in $(seq 1 $XCD); do run xcd_$i tests/test_gpu_xcd.py tests/test_gpu_xcd_outer.py tests/test_gpu_parity.py; done
cat "$OUT/tally.txt"
