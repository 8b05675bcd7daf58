"""GPU box: a short conic solve for `rocprofv3 --pmc ... -- python3 scripts/pmc_conic.py <c5|lasso> <direct|pcg> [max_admm_iters]` (counter passes
need few dispatches of the kernels they are after)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abip_amd import problems, qcp

what = sys.argv[1] if len(sys.argv) > 1 else "c5"
pcg = (sys.argv[2] if len(sys.argv) > 2 else "direct") == "pcg"
its = int(sys.argv[3]) if len(sys.argv) > 3 else 12
if what == "lasso":
    X, yv, lam = problems.lasso_protocol_data(5000, 15000)
    qcp.abip_ml(dict(X=X, y=yv, **{"lambda": lam}), dict(prob_type=0, eps=1e-3, linsys_solver=3 if pcg else 1, verbose=0, max_admm_iters=its))
else:
    data, K = problems.qcp_lasso_socp(10_000, 45_000)
    qcp.abip_qcp(data, K, dict(eps=1e-3, linsys_solver=3 if pcg else 1, verbose=0, max_admm_iters=its))
