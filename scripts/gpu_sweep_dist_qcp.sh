#!/bin/bash
# GPU: the conic column sharding over 2, 3 and 4 ranks on ONE GPU (host-staged gloo collective) on every case of tests/qcp_cases.py, against the
# single-GPU run of the same library: iteration counts, solution, bit-identical state across the ranks.  Usage: scripts/gpu_sweep_dist_qcp.sh [eps]
EPS=${1:-1e-5}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
for name in lasso_small mixed lp lasso_mid; do
  ref=$(PYTHONPATH=.:tests python - <<PY
from qcp_cases import make
from abip_amd import qcp
d, K = make("$name")
s, i = qcp.abip_qcp(d, K, dict(eps=$EPS, linsys_solver=3, verbose=0))
print(i["status"], i["ipm_iter"], i["admm_iter"], "%.10g" % i["pobj"])
PY
)
  for world in 2 3 4; do
    out=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node=$world --master-addr 127.0.0.1 --master-port $((29300 + world)) tests/dist_worker_qcp.py gloo-callback $name $EPS 2>/dev/null | grep '^RESULT ' | cut -c8- | python -c "import json,sys; o=json.load(sys.stdin); print(o['status'], o['ipm_iter'], o['admm_iter'], '%.10g' % o['pobj'], 'consistent' if o['consistent'] else 'INCONSISTENT', 'collectives', int(o['collectives']), 'cg/solve %.1f' % o['avg_cg_iters'])")
    echo "$name world $world: $out | single GPU: $ref"
  done
done
