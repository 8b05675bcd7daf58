#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats of the conic PCG back-end on C5, per-kernel totals -> gpurun_out/<tag>/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_prof_c5_pcg}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 "$ROOT/bench.py" --workload c5 --linsys indirect --no-cpu --no-to-tol > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
head -12 "$OUT/kernel_stats.csv" | cut -c1-200
rm -rf "$OUT/trace"
