#!/bin/bash
# the sharded solve with 4 and 8 ranks on ONE GPU, before the driver does it on eight.
#   (a) tests/dist_worker.py: world 4 / 8 x {gloo-callback, peer} x {rows, cols} on two small LPs
#   (b) bench.py --gpus 8 itself (it starts the ranks) under gloo-callback and peer, c3 size
#   (c) the same once at C4 size: wall time and peak host memory of 8 processes' set-up + device memory
# Output: gpurun_out/r06_eight/*.txt (summary in eight_rank_dry_run.txt)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_eight
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
SUM=$OUT/eight_rank_dry_run.txt
: > "$SUM"
port=29610
worker() {   # world mode fixture eps form
  port=$((port + 1))
  local tag="w$1_$2_$5_$(echo $3 | tr ':' '_')"
  local t0=$(date +%s%N)
  ABIP_HIP_DIST_CG=$5 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$1 --master-addr 127.0.0.1 --master-port $port \
      tests/dist_worker.py $2 $3 $4 > "$OUT/$tag.out" 2> "$OUT/$tag.err"
  local rc=$?
  local t1=$(date +%s%N)
  python3 - "$OUT/$tag.out" "$tag" $rc $(( (t1 - t0) / 1000000 ))ms >> "$SUM" <<'EOF'
import json, sys, hashlib
f, tag, rc, secs = sys.argv[1:5]
ln = [l for l in open(f) if l.startswith("RESULT ")]
if not ln:
    print(f"{tag}: rc={rc} NO RESULT ({secs})")
else:
    o = json.loads(ln[-1][7:])
    h = hashlib.sha256(repr((o["x"], o["y"], o["s"])).encode()).hexdigest()[:12]
    print(f"{tag}: rc={rc} status={o['status']} ipm={o['ipm_iter']} admm={o['admm_iter']} cg={o['cg']:.0f} cols={o['cols']:.0f} consistent={o['consistent']} xys_sha={h} ({secs})")
EOF
}
if [ "${1:-all}" = "all" ] || [ "$1" = "workers" ]; then
for W in 4 8; do
  for form in rows cols; do
    for mode in gloo-callback peer; do
      worker $W $mode lp_random_sparse_small 1e-06 $form
      worker $W $mode gen:skew:11 1e-05 $form
    done
  done
done
fi
bench() {   # tag transport args...
  local tag=$1 tr=$2; shift 2
  local t0=$(date +%s%N)
  ABIP_BENCH_TRANSPORT=$tr timeout 1500 python bench.py "$@" > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  local rc=$?
  local t1=$(date +%s%N)
  python3 - "$OUT/$tag.json" "$tag" $rc $(( (t1 - t0) / 1000000 ))ms >> "$SUM" <<'EOF'
import json, sys, re
f, tag, rc, secs = sys.argv[1:5]
ln = [l for l in open(f) if l.startswith("{")]
if not ln:
    print(f"{tag}: rc={rc} NO LINE ({secs})")
else:
    r = json.loads(ln[-1])
    print(f"{tag}: rc={rc} n_gpus={r['n_gpus']} rccl_ranks={r.get('rccl_ranks')} dist_cg={r.get('dist_cg')} value={r['value']:.2f} it/s ms_per_step={r['ms_per_step']:.3f} "
          f"other_form={ {k: round(v['value'], 2) for k, v in r['extra'].items() if k.startswith('dist_') and isinstance(v, dict)} } rank_rows={r.get('rank_rows')} "
          f"wall={secs}")
EOF
}
if [ "${1:-all}" = "all" ] || [ "$1" = "bench" ]; then
bench b8_c3_gloo gloo-callback --gpus 8 --workload c3 --steps 6 --warmup 2 --no-to-tol --no-cpu --no-extra
bench b8_c3_peer peer --gpus 8 --workload c3 --steps 6 --warmup 2 --no-to-tol --no-cpu --no-extra
bench b4_c3_peer peer --gpus 4 --workload c3 --steps 6 --warmup 2 --no-to-tol --no-cpu --no-extra
fi
if [ "${1:-all}" = "all" ] || [ "$1" = "c4" ]; then
# C4 size: memory sampled while the ranks run
( while true; do echo "$(date +%s) $(free -m | awk '/Mem:/{print $3}') MB host used; $(ps -C python -o rss= | awk '{s+=$1} END {printf "%d", s/1024}') MB python rss; $(rocm-smi --showmeminfo vram 2>/dev/null | awk '/Used/{print $NF}' | head -1) B vram used"; sleep 2; done ) > "$OUT/c4_mem_samples.txt" 2>&1 &
SAMPLER=$!
bench b8_c4_gloo gloo-callback --gpus 8 --workload c4 --steps 6 --warmup 2 --no-to-tol --no-cpu --no-extra
bench b8_c4_peer peer --gpus 8 --workload c4 --steps 6 --warmup 2 --no-to-tol --no-cpu --no-extra
kill $SAMPLER
python3 - "$OUT/c4_mem_samples.txt" >> "$SUM" <<'EOF'
import sys, re
host, vram, rss = [], [], []
for l in open(sys.argv[1]):
    m = re.match(r"\d+ (\d+) MB host used; (\d+) MB python rss; (\d*)", l)
    if m:
        host.append(int(m.group(1))); rss.append(int(m.group(2)))
        if m.group(3): vram.append(int(m.group(3)))
print(f"C4, 8 ranks on one box: host memory used min {min(host)} / max {max(host)} MB; resident set of all python processes max {max(rss)} MB; vram used max {max(vram) / 2**20 if vram else float('nan'):.0f} MiB")
EOF
fi
cat "$SUM"
