#!/bin/bash
# Round 6, VERDICT r5 item 6: the restart sums and running sums of the owned elements in registers for the whole persistent launch (developer build -DXCD_RSUM:
# make -C abip_amd/csrc exp EXPNAME=rsum EXPDEF=-DXCD_RSUM) against the default (four read-modify-write streams per element and iteration).  Output: gpurun_out/r06_rsum/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_rsum
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2 3; do
  for lib in default rsum; do
    if [ $lib = default ]; then unset ABIP_HIP_LIBRARY; else export ABIP_HIP_LIBRARY=$ROOT/abip_amd/lib/libabip_hip_rsum.so; fi
    for wl in c2 c3; do
      [ $wl = c3 ] && [ $rep != 1 ] && continue
      timeout 600 python bench.py --workload $wl --no-cpu --no-extra > "$OUT/${wl}_${lib}_$rep.json" 2> "$OUT/${wl}_${lib}_$rep.err"
      python3 - "$OUT/${wl}_${lib}_$rep.json" "$wl $lib $rep" <<'PY'
import json, sys
ln = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not ln: print(sys.argv[2], "NO LINE"); sys.exit(0)
r = json.loads(ln[-1]); tt = r["time_to_tol"]; ro = r["roofline"]
print(f"[{sys.argv[2]}]: window {r['value']:.1f} it/s ({ro['us_per_iteration']:.3f} us per iteration, {ro['exchanges_per_iteration']:.2f} exchanges of {ro['us_per_exchange']:.3f} us); whole solve {tt['status']} {tt['ipm_iter']}/{tt['admm_iter']}, {tt['solve_s']:.4f} s; res_pri {tt['res_pri']:.6e} gap {tt['rel_gap']:.6e}")
PY
    done
  done
done
