#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats of the conic direct back-end on C5 (set-up + solve), per-kernel totals -> gpurun_out/<tag>/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_prof_c5}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 "$ROOT/scripts/gpu_c5.py" 3 > "$OUT/c5_under_rocprof.txt" 2> "$OUT/c5_under_rocprof.err"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
d = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void abip::", "")
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out + "/kernel_totals.txt", "w") as fh:
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
        fh.write("%-60s launches %6d  total %10.1f us  mean %8.2f  max %9.2f\n" % (k[:60], len(v), sum(v), sum(v) / len(v), max(v)))
PY
head -25 "$OUT/kernel_totals.txt"; grep "^p=" "$OUT/c5_under_rocprof.txt"
rm -rf "$OUT/trace"
