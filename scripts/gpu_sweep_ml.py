"""GPU: randomized parity sweep of the specialised formulations (abip_ml surface: LASSO prob_type 0, SVM-SOCP 1, SVM-QP 3) --
seeded data of varied shapes, densities and regularisation weights (so that every branch of the formulations' scaling tables is
visited), device vs the oracle's restatement: status, outer and inner iteration counts, returned solution."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import __graft_entry__ as g
g.build()
from abip_amd import qcp
from oracle import pyoracle_qcp as pq

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
LS = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # device back-end: 1 direct, 3 PCG (the oracle always runs its reduced direct system)
bad = 0
rel = lambda a, r: np.linalg.norm(np.atleast_1d(a) - np.atleast_1d(r)) / max(np.linalg.norm(np.atleast_1d(r)), 1e-12)
for t in range(N):
    kind = (0, 1, 3)[t % 3]
    shape = t % 4
    if shape == 0: m, n = int(rng.integers(20, 120)), int(rng.integers(150, 600))        # wide
    elif shape == 1: m, n = int(rng.integers(200, 700)), int(rng.integers(5, 60))        # tall (m > 10 n sometimes, n < 10 sometimes)
    elif shape == 2: m, n = int(rng.integers(50, 300)), int(rng.integers(50, 300))       # squarish
    else: m, n = int(rng.integers(5, 30)), int(rng.integers(300, 900))                   # very wide (10 m < n)
    dens = float(rng.choice([0.02, 0.06, 0.15, 0.5, 1.0]))
    dens = max(dens, 3.0 / min(m, n))
    X = sp.random(m, n, density=min(1.0, dens), random_state=rng, data_rvs=rng.standard_normal, format="csc")
    eps = float(rng.choice([1e-3, 1e-5]))
    es = dict(eps=eps, eps_p=eps, eps_d=eps, eps_g=eps, eps_inf=eps, eps_unb=eps)
    t0 = time.time()
    if kind == 0:
        v = np.where(rng.random(n) < 0.5, rng.standard_normal(n) / np.sqrt(n), 0.0)
        y = X @ v + float(rng.choice([0.01, 1.0])) * rng.standard_normal(m)
        lam = float(np.abs(X.T @ y).max() / float(rng.choice([2, 5, 20])))
        want, wi = pq.solve_lasso(X, y, lam, max_ipm_iters=60, max_admm_iters=50000, **es)
        want = (want,)
        tag = f"lasso {m}x{n} d={dens:.2f} lam={lam:.2e}"
    else:
        y = np.sign(X @ rng.standard_normal(n) + 0.3 * rng.standard_normal(m)); y[y == 0] = 1.0
        lam = float(rng.choice([1e-3, 1e-2, 0.3, 2.0]))
        if kind == 1:
            lam = 1.0 / (m * lam) if rng.random() < 0.7 else lam       # C as test_svm.m:95, or raw weights on both sides of 1
            cap = 50 if np.any(np.diff(X.indptr) == 0) else 50000
            w0, b0, xi0, wi = pq.solve_svm(X, y, lam, max_ipm_iters=60, max_admm_iters=cap, **es)
        else:
            w0, b0, xi0, wi = pq.solve_svmqp(X, y, lam, max_ipm_iters=60, max_admm_iters=50000, **es)
        want = (w0, b0, xi0)
        tag = f"{'svm-socp' if kind == 1 else 'svm-qp  '} {m}x{n} d={dens:.2f} lam={lam:.2e}"
    tcpu = time.time() - t0
    sol, gi = qcp.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=kind, eps=eps, linsys_solver=LS, verbose=0, max_ipm_iters=60, max_admm_iters=50000))
    got = (sol["x"],) if kind == 0 else (sol["w"], sol["b"], sol["xi"])
    solved = wi["status_val"] in (1, 2)
    ex = max(rel(a, r) for a, r in zip(got, want)) if solved and all(np.all(np.isfinite(np.atleast_1d(r))) for r in want) else 0.0
    same_outer = gi["ipm_iter"] == wi["ipm_iter"] or (LS == 3 and abs(gi["ipm_iter"] - wi["ipm_iter"]) <= 1)
    zero_col = kind == 1 and bool(np.any(np.diff(X.indptr) == 0))   # the SVM-SOCP scaling divides by column norms: the oracle (like the reference) iterates on NaN, the device refuses at set-up
    if zero_col:
        print(f"--  {tag:46s} all-zero feature column: device status {gi['status']} (expected Failure), oracle {wi['status']} after {wi['admm_iter']} iterations", flush=True)
        bad += gi["status"] != "Failure"
        continue
    if LS == 3:   # inexact KKT solves on the device against the oracle's exact ones: same optimum, not the same trajectory -- compare the objectives
        dp = abs(gi["pobj"] - wi["pobj"]) / max(1.0, abs(wi["pobj"]))
        ok = gi["status"] == wi["status"] and dp <= 100 * eps and abs(gi["ipm_iter"] - wi["ipm_iter"]) <= 2
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {tag:46s} eps {eps:.0e} status {gi['status']}/{wi['status']} admm {gi['admm_iter']}/{wi['admm_iter']} ipm {gi['ipm_iter']}/{wi['ipm_iter']} |dpobj| {dp:.1e} cg/solve {gi['avg_cg_iters']:.1f}", flush=True)
        continue
    ok = gi["status"] == wi["status"] and same_outer and (gi["ipm_iter"] != wi["ipm_iter"] or abs(gi["admm_iter"] - wi["admm_iter"]) <= 0.03 * wi["admm_iter"] + 3) and ex < max(1e-3, 30 * eps)
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {tag:46s} eps {eps:.0e} status {gi['status']}/{wi['status']} admm {gi['admm_iter']}/{wi['admm_iter']} ipm {gi['ipm_iter']}/{wi['ipm_iter']} rel {ex:.1e} cpu {tcpu:.1f}s", flush=True)
print("FAILURES", bad)
