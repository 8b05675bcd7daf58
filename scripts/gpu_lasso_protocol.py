"""The reference's LASSO protocol (scripts/bench-qcp/test_lasso.m:39-91, get_lasso_simu_data.m:3-14: sprandn density 0.15,
v_i ~ N(0, 1/n) w.p. 1/2, y = X v + N(0, 1) noise, lambda = |X'y|_inf / 5, eps 1e-3) through the LASSO front end
(abip_ml, prob_type 0) on the device, both KKT back-ends, and through the generic conic path for comparison."""
import sys
import time

import numpy as np
import scipy.sparse as sp

from abip_amd import qcp


def data(m, n, seed=1, density=0.15):
    rng = np.random.default_rng(seed)
    X = sp.random(m, n, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    v = np.where(rng.random(n) > 0.5, rng.standard_normal(n) / n, 0.0)
    y = X @ v + rng.standard_normal(m)
    return X, y, float(np.abs(X.T @ y).max() / 5)


def socp(X, y, lam):
    p, d = X.shape
    r1 = sp.hstack([sp.csc_matrix(np.array([[1.0, -1.0]])), sp.csc_matrix((1, p + 2 * d))])
    r2 = sp.hstack([sp.csc_matrix((p, 2)), sp.identity(p), -X, X])
    A = sp.vstack([r1, r2]).tocsc(); A.sort_indices()
    return dict(A=A, b=np.concatenate([[1.0], -y]), c=np.concatenate([[0.5, 0.5], np.zeros(p), lam * np.ones(2 * d)])), dict(q=[p + 2], l=2 * d)


sizes = [(1000, 5000), (2000, 10000), (5000, 15000)] if len(sys.argv) < 2 else [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
for (m, n) in sizes:
    X, y, lam = data(m, n)
    f = lambda b: 0.5 * float(np.sum((X @ b - y) ** 2)) + lam * float(np.abs(b).sum())
    for ls in (1, 3):
        t = time.time(); sol, info = qcp.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, eps=1e-3, linsys_solver=ls, verbose=0)); t = time.time() - t
        print(f"lasso {m}x{n} front-end ls={ls}: {info['status']} ipm {info['ipm_iter']} admm {info['admm_iter']} setup {info['setup_time']:.3f}s solve {info['solve_time']:.3f}s "
              f"pobj {info['pobj']:.6f} f(beta) {f(sol['x']):.6f} cg {info['avg_cg_iters']:.1f}", flush=True)
    dq, K = socp(X, y, lam)
    for ls in (1, 3):
        sol, info = qcp.abip_qcp(dq, K, dict(eps=1e-3, linsys_solver=ls, verbose=0))
        beta = sol["x"][m + 2:m + 2 + n] - sol["x"][m + 2 + n:]
        print(f"lasso {m}x{n} generic   ls={ls}: {info['status']} ipm {info['ipm_iter']} admm {info['admm_iter']} setup {info['setup_time']:.3f}s solve {info['solve_time']:.3f}s "
              f"pobj {info['pobj']:.6f} f(beta) {f(beta):.6f} cg {info['avg_cg_iters']:.1f}", flush=True)
