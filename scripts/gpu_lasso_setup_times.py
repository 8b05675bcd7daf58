"""Where the direct back-end's set-up time goes on the reference's largest LASSO size (ABIP_HIP_SETUP_TIMES=1 prints the host phases)."""
import os, sys, time
os.environ["ABIP_HIP_SETUP_TIMES"] = "1"
from abip_amd import problems, qcp
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 15000)
X, y, lam = problems.lasso_protocol_data(m, n)
t = time.time()
sol, info = qcp.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, eps=1e-3, linsys_solver=1, verbose=0))
print(f"lasso {m}x{n} direct: {info['status']} admm {info['admm_iter']} setup {info['setup_time']:.3f}s solve {info['solve_time']:.3f}s wall {time.time() - t:.3f}s factor {info['factor']}")
