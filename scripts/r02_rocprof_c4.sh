#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats of the default bench workload (C4), summaries -> gpurun_out/<tag>/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_prof}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 "$ROOT/bench.py" --no-extra --no-cpu --no-to-tol > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# medians of the launches that did work (>= 8 us) for the PCG kernels, from the trace
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics, collections
out = sys.argv[1]
d = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void abip::", "")
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out + "/kernel_medians.txt", "w") as fh:
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        w = [x for x in v if x >= 8.0] or v
        fh.write("%-60s launches %6d  total %9.1f us  median(all) %7.2f  median(working, >= 8 us) %7.2f  n_working %d\n" % (k[:60], len(v), sum(v), statistics.median(v), statistics.median(w), len(w)))
PY
head -12 "$OUT/kernel_medians.txt"
