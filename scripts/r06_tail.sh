#!/bin/bash
# Round 6, VERDICT r5 item 4 (dense-tail stream kernels) and item 2 (C4 gaps outside the profiler): probes and A/B runs.  Output: gpurun_out/r06_tail/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_tail
mkdir -p "$OUT"
cd "$ROOT"
what=${1:-all}
if [ "$what" = all ] || [ "$what" = probe ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -o tools/tail_sym_probe tools/tail_sym_probe.hip 2> "$OUT/probe_build.err"
  timeout 300 tools/tail_sym_probe 10048 30 > "$OUT/tail_sym_probe_T10048.txt" 2>&1
  timeout 300 tools/tail_sym_probe 8960 30 > "$OUT/tail_sym_probe_T8960.txt" 2>&1
  timeout 300 tools/tail_sym_probe 2048 30 > "$OUT/tail_sym_probe_T2048.txt" 2>&1
  cat "$OUT/tail_sym_probe_T10048.txt"
fi
c5() {  # tag env...
  local tag=$1; shift
  env "$@" timeout 900 python bench.py --workload c5 --linsys direct --no-cpu > "$OUT/c5_$tag.json" 2> "$OUT/c5_$tag.err"
  python3 - "$OUT/c5_$tag.json" "$tag" <<'PY'
import json, sys
ln = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not ln: print(sys.argv[2], "NO LINE"); sys.exit(0)
r = json.loads(ln[-1]); ro = r["roofline"]
print(f"c5 direct [{sys.argv[2]}]: {r['value']:.0f} it/s, {r['steps']} its, solve {ro['avg_launch_us']:.1f} us, frac {ro['frac']:.3f}, status {r['time_to_tol']['status']}, pobj {r['extra']['pobj']:.12g}, setup {r['time_to_tol']['setup_s']:.2f} s")
PY
}
if [ "$what" = all ] || [ "$what" = c5 ]; then
  for rep in 1 2 3; do
    c5 waves2048_$rep ABIP_HIP_TAIL_WAVES=2048
    c5 waves1024_$rep ABIP_HIP_TAIL_WAVES=1024
    c5 waves1280_$rep ABIP_HIP_TAIL_WAVES=1280
  done
  c5 tri_lds_off ABIP_HIP_TRI_LDS=0
  c5 two_matvecs ABIP_HIP_TAIL_SYM=0
  for v in 1 0; do
    ABIP_HIP_TRI_LDS=$v timeout 900 python bench.py --workload c3 --linsys direct --no-cpu --no-extra > "$OUT/c3_direct_trilds$v.json" 2> "$OUT/c3_direct_trilds$v.err"
    python3 -c "
import json
r=json.loads([l for l in open('$OUT/c3_direct_trilds$v.json') if l.startswith('{')][-1]); tt=r['time_to_tol']
print('c3 --linsys direct [ABIP_HIP_TRI_LDS=$v]:', round(r['value'],1), 'it/s; whole solve', tt['status'], tt['ipm_iter'], tt['admm_iter'], round(tt['solve_s'],3), 's; solve', round(r['roofline']['avg_launch_us'],1), 'us, tail', r['roofline']['dense_tail'])
"
  done
fi
if [ "$what" = all ] || [ "$what" = tests ]; then
  timeout 1500 python -m pytest tests/test_gpu_qdldl_pin.py tests/test_gpu_qcp.py tests/test_gpu_baseline_size.py -x -q -m gpu --durations=15 > "$OUT/pytest_tail.txt" 2>&1; tail -25 "$OUT/pytest_tail.txt"
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "direct or tail or ldl" > "$OUT/pytest_parity_direct.txt" 2>&1; tail -5 "$OUT/pytest_parity_direct.txt"
fi
if [ "$what" = all ] || [ "$what" = gaps ]; then
  rm -f /tmp/st.txt
  ABIP_HIP_STAMP_DUMP=/tmp/st.txt timeout 900 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu --no-extra --no-to-tol > "$OUT/c4_gaps_bench.json" 2> "$OUT/c4_gaps_bench.err"
  python3 scripts/stamp_gaps.py /tmp/st.txt > "$OUT/c4_stamp_gaps_steps20.txt" 2>&1
  cp /tmp/st.txt "$OUT/c4_stamps_raw_steps20.txt"
  rm -f /tmp/st.txt
  ABIP_HIP_STAMP_DUMP=/tmp/st.txt timeout 900 python bench.py --workload c4 --steps 200 --warmup 20 --no-cpu --no-extra --no-to-tol > "$OUT/c4_gaps_bench200.json" 2>> "$OUT/c4_gaps_bench.err"
  python3 scripts/stamp_gaps.py /tmp/st.txt > "$OUT/c4_stamp_gaps_steps200.txt" 2>&1
  cat "$OUT/c4_stamp_gaps_steps20.txt" "$OUT/c4_stamp_gaps_steps200.txt"
  python3 -c "
import json
for f in ('c4_gaps_bench.json','c4_gaps_bench200.json'):
    r=json.loads([l for l in open('$OUT/'+f) if l.startswith('{')][-1]); print(f, r['value'], r['ms_per_step'], r['extra']['cg_iters_per_step'], r['extra']['cg_iters_executed_per_step'])
"
fi
if [ "$what" = kernarg ]; then   # where the kernel arguments live: does the ~4.8 us launch boundary move?
  for v in 0 1; do
    rm -f /tmp/st.txt
    HIP_FORCE_DEV_KERNARG=$v ABIP_HIP_STAMP_DUMP=/tmp/st.txt timeout 900 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu --no-extra --no-to-tol > "$OUT/c4_kernarg$v.json" 2> "$OUT/c4_kernarg$v.err"
    echo "HIP_FORCE_DEV_KERNARG=$v: $(python3 -c "import json; r=json.loads([l for l in open('$OUT/c4_kernarg$v.json') if l.startswith('{')][-1]); print(round(r['value'],1), 'it/s', round(r['ms_per_step'],3), 'ms/step')")"
    python3 scripts/stamp_gaps.py /tmp/st.txt | grep -A1 "one boundary\|two boundaries" | cut -c1-200
  done
fi
