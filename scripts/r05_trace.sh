#!/bin/bash
# GPU box: rocprofv3 kernel traces of the bench windows the round-5 records are quoted on (LP: c4 / c2 / c3; conic: c5 direct, c5 PCG, the LASSO protocol), the
# per-kernel medians, the step breakdown of the C4 window, and the durations file bench.py reads (trace / stamp ratio + a hash of the kernel sources it was taken on).
#   usage: [WLS="c2 c3"] scripts/r05_trace.sh [tag]  ->  gpurun_out/<tag>/{c4,c2,c3,c5_direct,c5_pcg,lasso}_kernel_stats.csv, *_kernel_medians.txt, trace_durations.json, the bench lines printed under the profiler
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05_trace}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for wl in ${WLS:-c4 c2 c3 c5_direct c5_pcg lasso}; do
  case $wl in
    c4) args="--workload c4 --steps 20 --warmup 5" ;;      # the driver's window
    c2|c3) args="--workload $wl" ;;                        # the windows their records on the default line are quoted on
    c5_direct) args="--workload c5 --linsys direct" ;;
    c5_pcg) args="--workload c5 --linsys indirect" ;;
    lasso) args="--workload lasso" ;;
  esac
  cmd="rocprofv3 --kernel-trace --stats -- python3 bench.py $args --no-cpu --no-extra --no-to-tol"
  timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/$wl" --output-format csv -- python3 "$ROOT/bench.py" $args --no-cpu --no-extra --no-to-tol \
      > "$OUT/${wl}_bench_under_rocprof.json" 2> "$OUT/${wl}.err"
  echo "$wl rc $?" >> "$OUT/summary.txt"
  f=$(find "$OUT/$wl" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${wl}_kernel_stats.csv"
  t=$(find "$OUT/$wl" -name '*kernel_trace.csv' | head -1)
  if [ -n "$t" ]; then
    if [ "$wl" = c4 ]; then
      python3 "$ROOT/scripts/trace_medians.py" "$t" --json "$OUT/trace_durations.json" --bench "$OUT/c4_bench_under_rocprof.json" --cmd "$cmd" > "$OUT/c4_kernel_medians.txt" 2>> "$OUT/c4.err"
      python3 "$ROOT/scripts/step_breakdown.py" "$t" > "$OUT/c4_step_breakdown.txt" 2>&1
    else
      python3 "$ROOT/scripts/trace_medians.py" "$t" --cmd "$cmd" > "$OUT/${wl}_kernel_medians.txt" 2>> "$OUT/${wl}.err"
    fi
  fi
  find "$OUT/$wl" -name '*kernel_trace.csv' -size +20M -delete
done
cat "$OUT/summary.txt"
