#!/bin/bash
# Round 6, VERDICT r5 item 5: what the halo form of the persistent launch's PCG would ADD to the first exchange of a PCG iteration, measured: the developer build
# with -DXCD_HALO_PROBE (make -C abip_amd/csrc exp EXPNAME=haloprobe EXPDEF=-DXCD_HALO_PROBE) gathers and row-sums two more slices of the same size there.
# Same trajectory, same counts; the difference in time per PCG iteration is the price.   Output: gpurun_out/r06_halo/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_halo
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2 3; do
  for lib in default haloprobe; do
    if [ $lib = default ]; then unset ABIP_HIP_LIBRARY; else export ABIP_HIP_LIBRARY=$ROOT/abip_amd/lib/libabip_hip_haloprobe.so; fi
    timeout 600 python bench.py --workload c3 --no-cpu --no-extra > "$OUT/c3_${lib}_$rep.json" 2> "$OUT/c3_${lib}_$rep.err"
    python3 - "$OUT/c3_${lib}_$rep.json" "$lib $rep" <<'PY'
import json, sys
ln = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not ln: print(sys.argv[2], "NO LINE"); sys.exit(0)
r = json.loads(ln[-1]); tt = r["time_to_tol"]; ro = r["roofline"]
cg = tt["cg_iters_executed_per_step"] * tt["admm_iter"]
print(f"c3 [{sys.argv[2]}]: window {r['value']:.1f} it/s; whole solve {tt['status']} {tt['ipm_iter']}/{tt['admm_iter']}, {tt['solve_s']:.3f} s, executed PCG iterations {cg:.0f} -> {1e6 * tt['solve_s'] / cg:.3f} us per PCG iteration (everything else included); us per exchange {ro['us_per_exchange']:.3f}, exchanges per iteration {ro['exchanges_per_iteration']:.1f}")
PY
  done
done
