#!/bin/bash
# c4 only: refresh the driver-window trace on the final tree (same recipe as scripts/r05_trace.sh)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05r_trace
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
args="--workload c4 --steps 20 --warmup 5"
cmd="rocprofv3 --kernel-trace --stats -- python3 bench.py $args --no-cpu --no-extra --no-to-tol"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/c4" --output-format csv -- python3 "$ROOT/bench.py" $args --no-cpu --no-extra --no-to-tol > "$OUT/c4_bench_under_rocprof.json" 2> "$OUT/c4.err"
f=$(find "$OUT/c4" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/c4_kernel_stats.csv"
t=$(find "$OUT/c4" -name '*kernel_trace.csv' | head -1)
python3 "$ROOT/scripts/trace_medians.py" "$t" --json "$OUT/trace_durations.json" --bench "$OUT/c4_bench_under_rocprof.json" --cmd "$cmd" > "$OUT/c4_kernel_medians.txt" 2>> "$OUT/c4.err"
python3 "$ROOT/scripts/step_breakdown.py" "$t" > "$OUT/c4_step_breakdown.txt" 2>&1
find "$OUT/c4" -name '*kernel_trace.csv' -size +20M -delete
head -8 "$OUT/c4_step_breakdown.txt"
