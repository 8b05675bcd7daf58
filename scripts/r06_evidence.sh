#!/bin/bash
# Round 6 evidence on one GPU box: kernel traces of every bench window (scripts/r05_trace.sh), HBM-traffic counter passes (scripts/r05_pmc.sh), then -- with the
# fresh trace / stamp ratio in place -- the driver's window, the default line and smoke().  Output: gpurun_out/r06_trace, r06_pmc, r06_lines.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r06_lines
WLS="${WLS:-c4 c2 c3 c5_direct c5_pcg lasso}" bash scripts/r05_trace.sh r06_trace > gpurun_out/r06_lines/trace.log 2>&1
[ -f gpurun_out/r06_trace/trace_durations.json ] && cp gpurun_out/r06_trace/trace_durations.json profiles/r06_trace_durations.json
CASES="${CASES:-c4 c5_direct c2 c3}" bash scripts/r05_pmc.sh r06_pmc > gpurun_out/r06_lines/pmc.log 2>&1
[ -s gpurun_out/r06_pmc/pmc_traffic.json ] && cp gpurun_out/r06_pmc/pmc_traffic.json profiles/r06_pmc_traffic.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_lines/driver_window_bench_line.json 2> gpurun_out/r06_lines/driver_window.err
python bench.py > gpurun_out/r06_lines/default_bench_line.json 2> gpurun_out/r06_lines/default.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r06_lines/smoke.txt 2>&1
cp profiles/r06_trace_durations.json profiles/r06_pmc_traffic.json gpurun_out/r06_lines/ 2>/dev/null
python3 - <<'PY'
import json
for f in ("driver_window_bench_line.json", "default_bench_line.json"):
    try:
        r = json.loads([l for l in open("gpurun_out/r06_lines/" + f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    ro = r["roofline"]; tt = r.get("time_to_tol") or {}
    print(f"{f}: {r['value']:.1f} it/s ({r['steps']} steps), whole solve {r.get('value_whole_solve')}, to tol {tt.get('solve_s')} s {tt.get('status')} {tt.get('ipm_iter')}/{tt.get('admm_iter')}; roofline frac {ro['frac']:.3f} ({ro['avg_launch_us']:.2f} us, stale={ro.get('trace_file_stale')}), traffic {ro.get('traffic')}; cpu {r['cpu_baseline']['value'] if r.get('cpu_baseline') else None} rel_err {r['cpu_baseline'].get('rel_err_xys') if r.get('cpu_baseline') else None}")
    for k, v in (r["extra"].get("configs") or {}).items():
        if "error" in v: print("   ", k, v); continue
        t2 = v.get("time_to_tol") or {}
        print(f"    {k}: {v['value']:.1f} it/s, whole {v.get('value_whole_solve')}, to tol {t2.get('solve_s')} s ({t2.get('status')}), frac {v['roofline']['frac']:.3f}, us {v['roofline'].get('avg_launch_us')}")
PY
tail -3 gpurun_out/r06_lines/smoke.txt
