"""Seeded conic problems shared by the multi-rank worker (tests/dist_worker_qcp.py) and its test."""
import numpy as np
import scipy.sparse as sp


def lasso_socp(p, dft, seed, density=0.2):
    rng = np.random.default_rng(seed)
    X = sp.random(p, dft, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    yv = X @ (rng.standard_normal(dft) * (rng.random(dft) < 0.3)) + 0.01 * rng.standard_normal(p)
    lam = np.abs(X.T @ yv).max() / 5
    r1 = sp.hstack([sp.csc_matrix(np.array([[1.0, -1.0]])), sp.csc_matrix((1, p + 2 * dft))])
    r2 = sp.hstack([sp.csc_matrix((p, 2)), sp.identity(p), -X, X])
    A = sp.vstack([r1, r2]).tocsc()
    b = np.concatenate([[1.0], -yv]); c = np.concatenate([[0.5, 0.5], np.zeros(p), lam * np.ones(2 * dft)])
    return dict(A=A, b=b, c=c), dict(q=[p + 2], l=2 * dft)


def mixed(seed):
    """several SOCs and rotated cones, free, zero and orthant blocks, and a diagonal Q"""
    rng = np.random.default_rng(seed)
    sizes_q, sizes_rq, f, zc, l = [7, 12, 1, 30], [5, 9], 4, 2, 60
    n2 = sum(sizes_q) + sum(sizes_rq) + f + zc + l
    m2 = 25
    A2 = sp.random(m2, n2, density=0.25, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    x0 = np.zeros(n2); pos = 0
    for sz in sizes_q:
        v = rng.standard_normal(sz); v[0] = np.linalg.norm(v[1:]) + 1.0; x0[pos:pos + sz] = v; pos += sz
    for sz in sizes_rq:
        v = rng.standard_normal(sz); v[0] = 1.0 + abs(v[0]); v[1] = (v[2:] @ v[2:]) / (2 * v[0]) + 0.5; x0[pos:pos + sz] = v; pos += sz
    x0[pos:pos + f] = rng.standard_normal(f); pos += f + zc
    x0[pos:] = rng.random(l) + 0.1
    nc = sum(sizes_q) + sum(sizes_rq)
    c = A2.T @ rng.standard_normal(m2) + np.concatenate([x0[:nc], np.zeros(f), rng.standard_normal(zc), rng.random(l) + 0.1])
    Q = sp.diags(rng.random(n2) * (rng.random(n2) < 0.5)).tocsc()
    return dict(A=A2, b=A2 @ x0, c=c, Q=Q), dict(q=sizes_q, rq=sizes_rq, f=f, z=zc, l=l)


def make(name):
    if name == "lasso_small":
        return lasso_socp(40, 120, 3)
    if name == "lasso_mid":
        return lasso_socp(300, 2000, 5, density=0.05)
    if name == "mixed":
        return mixed(21)
    if name == "lp":
        rng = np.random.default_rng(9)
        m, n = 30, 90
        A = sp.hstack([sp.identity(m), sp.random(m, n - m, density=0.2, random_state=rng, data_rvs=rng.standard_normal)]).tocsc()
        x0 = rng.random(n) * (rng.random(n) < 0.5) + np.concatenate([np.ones(m), np.zeros(n - m)])
        return dict(A=A, b=A @ x0, c=rng.random(n) + 0.1), dict(l=n)
    raise KeyError(name)
