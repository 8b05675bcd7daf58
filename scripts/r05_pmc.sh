#!/bin/bash
# GPU box: HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel-trace only -- MI355X_MICROARCH.md, HBM / rocprofv3) over
# short slices of C4 (LP PCG), C5 direct, C5 PCG and the LASSO protocol.   usage: [CASES="c2 c3"] scripts/r05_pmc.sh [tag]  ->  gpurun_out/<tag>/<case>_<counter>/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05_pmc}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() { # name, counter, program args...
  local name=$1 ctr=$2; shift 2
  case " ${CASES:-c4 c5_direct c5_pcg lasso_pcg c2 c3} " in *" $name "*) ;; *) return 0 ;; esac
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$OUT/${name}_$ctr" --output-format csv -- python3 "$@" > "$OUT/${name}_$ctr.log" 2>&1
  echo "$name $ctr: rc $?" >> "$OUT/summary.txt"
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  run c4 $ctr "$ROOT/scripts/pmc_c4.py" 6
  run c5_direct $ctr "$ROOT/scripts/pmc_conic.py" c5 direct 12
  run c5_pcg $ctr "$ROOT/scripts/pmc_conic.py" c5 pcg 12
  run lasso_pcg $ctr "$ROOT/scripts/pmc_conic.py" lasso pcg 8
  run c2 $ctr "$ROOT/scripts/pmc_lp.py" c2 600
  run c3 $ctr "$ROOT/scripts/pmc_lp.py" c3 60
done
python3 "$ROOT/scripts/pmc_summarize.py" "$OUT" > "$OUT/pmc_traffic.json" 2>> "$OUT/summary.txt"   # (copied to profiles/rNN_pmc_traffic.json)
cat "$OUT/summary.txt"
