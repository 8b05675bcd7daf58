"""GPU: the conic path on LASSO-as-SOCP at growing sizes up to config C5 (p = 10 000, d = 45 000)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from abip_amd import problems, qcp

for (p, d) in [(2000, 9000), (5000, 22500), (10000, 45000)][: int(sys.argv[1]) if len(sys.argv) > 1 else 3]:
    t0 = time.time()
    data, K = problems.qcp_lasso_socp(p, d)
    tg = time.time() - t0
    for eps in (1e-3,):
        t0 = time.time()
        sol, info = qcp.abip_qcp(data, K, dict(eps=eps, linsys_solver=1, verbose=0))
        wall = time.time() - t0
        beta = sol["x"][p + 2: p + 2 + d] - sol["x"][p + 2 + d:]
        X = -data["A"][1:, p + 2: p + 2 + d]
        yv = -data["b"][1:]
        obj = 0.5 * np.sum((X @ beta - yv) ** 2) + data["c"][-1] * np.abs(beta).sum()
        print(f"p={p} d={d} n={data['A'].shape[1]} nnz={data['A'].nnz} gen {tg:.1f}s eps={eps:g}: {info['status']} ipm {info['ipm_iter']} admm {info['admm_iter']} "
              f"setup {info['setup_time']:.2f}s solve {info['solve_time']:.2f}s wall {wall:.2f}s  it/s {info['admm_iter'] / max(info['solve_time'], 1e-9):.1f} "
              f"pobj {info['pobj']:.6f} lasso-obj {obj:.6f} res {info['res_pri']:.1e} {info['res_dual']:.1e} {info['gap']:.1e}", flush=True)
