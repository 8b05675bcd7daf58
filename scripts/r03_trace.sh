#!/bin/bash
# GPU box: rocprofv3 kernel traces of the bench windows the round-3 records are quoted on.
#   usage: scripts/r03_trace.sh [tag]  ->  gpurun_out/<tag>/{c4,c2,c3}/..._kernel_stats.csv + the bench line printed under the profiler
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03_trace}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for wl in c4 c2 c3; do
  win="--steps 20 --warmup 5"                       # c4: the driver's window; c2 / c3: the windows their records on the default line are quoted on
  [ "$wl" != c4 ] && win=""
  timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/$wl" --output-format csv -- python3 "$ROOT/bench.py" --workload $wl $win --no-cpu --no-extra --no-to-tol \
      > "$OUT/${wl}_bench_under_rocprof.json" 2> "$OUT/${wl}.err"
  echo "$wl rc $?" >> "$OUT/summary.txt"
  f=$(find "$OUT/$wl" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${wl}_kernel_stats.csv"
  t=$(find "$OUT/$wl" -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && [ "$wl" = c4 ] && python3 "$ROOT/scripts/step_breakdown.py" "$t" > "$OUT/c4_step_breakdown.txt" 2>&1
  find "$OUT/$wl" -name '*kernel_trace.csv' -size +20M -delete
done
cat "$OUT/summary.txt"
