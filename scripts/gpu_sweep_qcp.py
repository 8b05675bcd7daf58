"""GPU: randomized parity sweep of the conic path -- seeded mixed-cone SOCPs, LASSOs and QPs of varied shapes, device vs oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, scipy.sparse as sp
import __graft_entry__ as g
g.build()
from abip_amd import qcp
from oracle import pyoracle_qcp as pq
from test_gpu_qcp import lasso_socp

def mixed(rng):
    nq, nr = int(rng.integers(1, 6)), int(rng.integers(0, 4))
    sizes_q = [int(rng.integers(1, 40)) for _ in range(nq)]; sizes_rq = [int(rng.integers(3, 30)) for _ in range(nr)]
    f, zc, l = int(rng.integers(0, 5)), int(rng.integers(0, 3)), int(rng.integers(1, 40))
    n2 = sum(sizes_q) + sum(sizes_rq) + f + zc + l; m2 = int(rng.integers(3, max(4, n2 // 3)))
    A2 = sp.random(m2, n2, density=min(1.0, 6.0 / n2 + 0.15), random_state=rng, data_rvs=rng.standard_normal, format="csc")
    x0 = np.zeros(n2); pos = 0
    for sz in sizes_q:
        v = rng.standard_normal(sz); v[0] = np.linalg.norm(v[1:]) + 1.0; x0[pos:pos + sz] = v; pos += sz
    for sz in sizes_rq:
        v = rng.standard_normal(sz); v[0] = 1.0 + abs(v[0]); v[1] = (v[2:] @ v[2:]) / (2 * v[0]) + 0.5; x0[pos:pos + sz] = v; pos += sz
    x0[pos:pos + f] = rng.standard_normal(f); pos += f + zc
    x0[pos:] = rng.random(l) + 0.1
    nc = sum(sizes_q) + sum(sizes_rq)
    c = A2.T @ rng.standard_normal(m2) + np.concatenate([x0[:nc], np.zeros(f), rng.standard_normal(zc), rng.random(l) + 0.1])
    return dict(A=A2, b=A2 @ x0, c=c), dict(q=sizes_q, rq=sizes_rq, f=f, z=zc, l=l), None, f"mixed q{sizes_q} rq{sizes_rq} f{f} z{zc} l{l} m{m2}"

def qp(rng):
    m2, n2 = int(rng.integers(3, 20)), int(rng.integers(20, 80))
    A2 = sp.random(m2, n2, density=0.4, random_state=rng, format="csc") + sp.hstack([sp.identity(m2), sp.csc_matrix((m2, n2 - m2))])
    G = rng.standard_normal((n2, n2)); Q = sp.csc_matrix(G @ G.T / n2 + 0.1 * np.eye(n2))
    return dict(A=sp.csc_matrix(A2), b=A2 @ rng.random(n2), c=rng.standard_normal(n2), Q=Q), dict(l=n2), Q, f"qp {m2}x{n2}"

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
LS = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # 1: direct back-end, 3: the conic PCG back-end (QPs with a non-diagonal Q are skipped there)
for t in range(int(sys.argv[2]) if len(sys.argv) > 2 else 18):
    kind = t % 3
    if kind == 0: data, K, Q, tag = mixed(rng)
    elif kind == 1: data, K, Q, tag = qp(rng)
    else:
        p, d = int(rng.integers(20, 300)), int(rng.integers(50, 900)); data, K = lasso_socp(p, d, int(rng.integers(1, 10 ** 6)), density=min(1.0, 8.0 / p + 0.01)); Q = None; tag = f"lasso {p}x{d}"
    if LS == 3 and Q is not None and (Q - sp.diags(Q.diagonal())).nnz > 0:
        continue
    eps = 1e-5
    t0 = time.time()
    x, y, s, oi, _ = pq.solve(data["A"], data["b"], data["c"], K, Q=Q, eps=eps, eps_p=eps, eps_d=eps, eps_g=eps, eps_inf=eps, eps_unb=eps, linsys_solver=LS, max_admm_iters=20000, max_ipm_iters=60)
    tcpu = time.time() - t0
    sol, gi = qcp.abip_qcp(data, K, dict(eps=eps, linsys_solver=LS, verbose=0, max_admm_iters=20000, max_ipm_iters=60))
    rel = lambda a, r: np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-300)
    ex = max(rel(sol["x"], x), rel(sol["y"], y)) if oi["status_val"] in (1, 2) and np.all(np.isfinite(x)) else 0.0
    ok = gi["status"] == oi["status"] and abs(gi["ipm_iter"] - oi["ipm_iter"]) <= (1 if LS == 3 else 0) and (gi["ipm_iter"] != oi["ipm_iter"] or abs(gi["admm_iter"] - oi["admm_iter"]) <= 0.03 * oi["admm_iter"] + 3) and ex < 1e-3
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {tag[:58]:58s} status {gi['status']}/{oi['status']} admm {gi['admm_iter']}/{oi['admm_iter']} ipm {gi['ipm_iter']}/{oi['ipm_iter']} rel(xy) {ex:.1e} cpu {tcpu:.1f}s", flush=True)
print("FAILURES", bad)
