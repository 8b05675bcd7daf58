"""Seeded synthetic LP instances in ABIP standard form  min c'x  s.t. Ax = b, x >= 0.

No Netlib / Mittelmann files exist offline, so the bench and
the parity tests run on clearly-labelled structure-matched surrogates:

* ``lp_install_smoke``  -- the LP of the reference's only test
  (test/test_abip_install.m:7-21: ``A=[sprand(50,2000,0.3), speye(50)]``) with a
  documented PRNG instead of Matlab's ``rng(24)`` stream.
* ``lp_afiro_like``     -- 27 x 51, the shape of Netlib afiro (config C1).
* ``lp_staircase``      -- block-staircase LP, the structure class of Netlib
  25fv47 (config C2: m~821, n~1876, nnz~1.1e4).
* ``lp_multicommodity`` -- multi-commodity network flow with joint arc
  capacities, the structure class of Mittelmann/Kennington pds-xx (config C3).
* ``lp_random_sparse``  -- identity block + k random rows per column
  (config C4: m=200 000, n=500 000, 16 per column, nnz~5e6).

All generators use ``numpy.random.Generator(PCG64(seed))`` and return
``(A_csc, b, c)`` with ``A_csc`` a ``scipy.sparse.csc_matrix`` (sorted indices,
no explicit zeros, int64 index arrays).  Every instance is primal feasible
(b = A x0, x0 >= 0) and dual feasible (c = A'y0 + s0, s0 >= 0 or c > 0), hence
solvable.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = [
    "lp_install_smoke", "lp_afiro_like", "lp_staircase", "lp_multicommodity",
    "lp_random_sparse", "canonical_csc",
]


def canonical_csc(A) -> sp.csc_matrix:
    A = sp.csc_matrix(A, dtype=np.float64)
    A.sum_duplicates()
    A.eliminate_zeros()
    A.sort_indices()
    A.indices = A.indices.astype(np.int64)
    A.indptr = A.indptr.astype(np.int64)
    return A


def _finish(A, rng, x_density=0.3, dual=True):
    """b = A x0 with a sparse non-negative x0; c = A'y0 + s0 with s0 > 0 off supp(x0)."""
    A = canonical_csc(A)
    m, n = A.shape
    x0 = np.where(rng.random(n) < x_density, rng.random(n), 0.0)
    b = A @ x0
    if dual:
        y0 = rng.standard_normal(m)
        s0 = np.where(x0 > 0, 0.0, rng.random(n) + 0.1)
        c = A.T @ y0 + s0
    else:
        c = rng.random(n) + 0.1
    return A, np.ascontiguousarray(b), np.ascontiguousarray(c)


def lp_install_smoke(seed: int = 24, m: int = 50, n0: int = 2000, density: float = 0.3):
    rng = np.random.Generator(np.random.PCG64(seed))
    R = sp.random(m, n0, density=density, format="csc", random_state=rng, data_rvs=rng.random)
    A = canonical_csc(sp.hstack([R, sp.identity(m, format="csc")], format="csc"))
    b = rng.random(m)
    c = rng.random(n0 + m)
    return A, b, c


def lp_afiro_like(seed: int = 1):
    """27 x 51 with 32 structural columns (2-4 entries each) and 19 slack columns."""
    rng = np.random.Generator(np.random.PCG64(seed))
    m, ns, nslack = 27, 32, 19
    rows, cols, vals = [], [], []
    for j in range(ns):
        k = int(rng.integers(2, 5))
        r = rng.choice(m, size=k, replace=False)
        rows += list(r); cols += [j] * k; vals += list(rng.uniform(-2.0, 2.0, size=k).round(3) + 0.001)
    for t in range(nslack):
        rows.append(t); cols.append(ns + t); vals.append(1.0)
    # make sure every row is touched
    for i in range(m):
        if i not in rows:
            rows.append(i); cols.append(int(rng.integers(0, ns))); vals.append(1.0)
    A = sp.coo_matrix((vals, (rows, cols)), shape=(m, ns + nslack))
    return _finish(A, rng, x_density=0.5)


def lp_staircase(seed: int = 47, stages: int = 12, rows_per: int = 68, cols_per: int = 140,
                 diag_density: float = 0.07, link_density: float = 0.02, slack_frac: float = 0.25):
    """Block staircase: stage t couples its own columns (diagonal block) with stage t-1's
    (sub-diagonal linking block).  Defaults give m=816, n~1890, nnz~1.1e4 (25fv47 class)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    blocks = [[None] * stages for _ in range(stages)]
    for t in range(stages):
        Dg = sp.random(rows_per, cols_per, density=diag_density, format="csc", random_state=rng,
                       data_rvs=lambda k: rng.uniform(-1.0, 1.0, size=k) * 10.0 ** rng.integers(-1, 2, size=k))
        blocks[t][t] = Dg
        if t > 0:
            blocks[t][t - 1] = sp.random(rows_per, cols_per, density=link_density, format="csc", random_state=rng,
                                         data_rvs=lambda k: rng.uniform(-1.0, 1.0, size=k))
    S = sp.bmat(blocks, format="csc")
    m = S.shape[0]
    nsl = int(slack_frac * m)
    sl_rows = rng.choice(m, size=nsl, replace=False)
    Sl = sp.coo_matrix((np.ones(nsl), (sl_rows, np.arange(nsl))), shape=(m, nsl))
    # guarantee full row rank with a weak identity on rows no column touches
    A = sp.hstack([S, Sl], format="csc")
    empty = np.flatnonzero(np.diff(sp.csr_matrix(A).indptr) == 0)
    if empty.size:
        Ex = sp.coo_matrix((np.ones(empty.size), (empty, np.arange(empty.size))), shape=(m, empty.size))
        A = sp.hstack([A, Ex], format="csc")
    A = sp.csc_matrix(A)
    A = A[:, np.flatnonzero(np.diff(A.indptr) > 0)]              # no empty columns
    return _finish(A, rng, x_density=0.3)


def lp_multicommodity(seed: int = 10, nodes: int = 120, arcs: int = 520, commodities: int = 8):
    """min sum_k c_k'x_k  s.t.  N x_k = d_k (node balance, one redundant row dropped per
    commodity),  sum_k x_k + slack = cap  (joint capacity),  x, slack >= 0.
    pds-10 itself has 16 558 rows / 49 932 columns; scale nodes/arcs/commodities up for C3."""
    rng = np.random.Generator(np.random.PCG64(seed))
    # connected digraph: a ring plus random chords
    tails = list(range(nodes)); heads = [(i + 1) % nodes for i in range(nodes)]
    extra = arcs - nodes
    t2 = rng.integers(0, nodes, size=extra); h2 = rng.integers(0, nodes, size=extra)
    h2 = np.where(h2 == t2, (h2 + 1) % nodes, h2)
    tails = np.concatenate([np.array(tails), t2]); heads = np.concatenate([np.array(heads), h2])
    na = tails.size
    Ninc = sp.coo_matrix((np.concatenate([np.ones(na), -np.ones(na)]),
                          (np.concatenate([tails, heads]), np.concatenate([np.arange(na), np.arange(na)]))),
                         shape=(nodes, na)).tocsr()[:-1, :]          # drop one (redundant) balance row
    K = commodities
    top = sp.block_diag([Ninc] * K, format="csc")                   # (K*(nodes-1)) x (K*na)
    cap = sp.hstack([sp.identity(na, format="csc")] * K + [sp.identity(na, format="csc")], format="csc")
    top = sp.hstack([top, sp.csc_matrix((top.shape[0], na))], format="csc")
    A = canonical_csc(sp.vstack([top, cap], format="csc"))
    n = A.shape[1]
    # feasible flow: route random positive circulations + keep slack positive
    x0 = np.zeros(n)
    x0[: K * na] = np.where(rng.random(K * na) < 0.3, rng.random(K * na), 0.0)
    x0[K * na:] = rng.random(na) + 0.05
    b = A @ x0
    c = np.concatenate([rng.random(K * na) + 0.1, np.zeros(na)])
    return A, np.ascontiguousarray(b), np.ascontiguousarray(c)


def lp_random_sparse(m: int = 200_000, n: int = 500_000, per_col: int = 16, seed: int = 88172645463325252 % (2 ** 32),
                     x_density: float = 0.3):
    """Config C4: columns 0..m-1 are the identity (full row rank, feasible);
    every other column gets ``per_col`` uniformly random rows (duplicates merged) with U(-1,1)
    values; x0 sparse U(0,1) plus 1 on the identity block; b = A x0; c ~ U(0.1, 1.1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nr = n - m
    rows = rng.integers(0, m, size=(nr, per_col), dtype=np.int64)
    rows.sort(axis=1)
    vals = rng.uniform(-1.0, 1.0, size=(nr, per_col))
    dup = np.zeros_like(rows, dtype=bool)
    dup[:, 1:] = rows[:, 1:] == rows[:, :-1]
    keep = ~dup
    counts = keep.sum(axis=1)
    indptr = np.zeros(n + 1, dtype=np.int64)
    indptr[1: m + 1] = np.arange(1, m + 1)
    indptr[m + 1:] = m + np.cumsum(counts)
    indices = np.concatenate([np.arange(m, dtype=np.int64), rows[keep]])
    data = np.concatenate([np.ones(m), vals[keep]])
    A = sp.csc_matrix((data, indices, indptr), shape=(m, n))
    A.has_sorted_indices = True
    x0 = np.where(rng.random(n) < x_density, rng.random(n), 0.0)
    x0[:m] += 1.0
    b = A @ x0
    c = rng.uniform(0.1, 1.1, size=n)
    return A, np.ascontiguousarray(b), np.ascontiguousarray(c)


def lasso_data(p: int = 10_000, d: int = 45_000, density: float = 0.005, seed: int = 5):
    """(X, y, lambda) of config C5: scripts/bench-qcp/get_lasso_simu_data.m:3-14 (sparse Gaussian X, v_i ~ N(0, 1/d) w.p. 1/2,
    y = X v + noise, lambda = ||X'y||_inf / 5) -- what the LASSO front end (abip_ml, prob_type 0) takes as it stands."""
    rng = np.random.default_rng(seed)
    X = sp.random(p, d, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    v = rng.standard_normal(d) / np.sqrt(d) * (rng.random(d) < 0.5)
    y = X @ v + 0.01 * rng.standard_normal(p)
    return X, y, float(np.abs(X.T @ y).max() / 5)


def lasso_protocol_data(m: int = 5000, n: int = 15000, seed: int = 1, density: float = 0.15):
    """(X, y, lambda) of the reference's own LASSO benchmark (scripts/bench-qcp/test_lasso.m:39-44 sizes 1000..5000 x 5000..15000,
    get_lasso_simu_data.m:3-14): sprandn-like X of density 0.15, v_i ~ N(0, 1/n) w.p. 1/2, y = X v + N(0, 1), lambda = |X'y|_inf / 5.
    (Matlab's rng stream is not reproducible here: numpy default_rng(seed).)"""
    rng = np.random.default_rng(seed)
    X = sp.random(m, n, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    v = np.where(rng.random(n) > 0.5, rng.standard_normal(n) / n, 0.0)
    y = X @ v + rng.standard_normal(m)
    return X, y, float(np.abs(X.T @ y).max() / 5)


def qcp_lasso_socp(p: int = 10_000, d: int = 45_000, density: float = 0.005, seed: int = 5):
    """Config C5: LASSO  min 1/2 ||X beta - y||^2 + lam ||beta||_1  as an SOCP for the conic path.
    Data as scripts/bench-qcp/get_lasso_simu_data.m:3-14 (sparse Gaussian X, v_i ~ N(0, 1/d) w.p. 1/2, y = X v + noise,
    lam = ||X'y||_inf / 5).  Variables (q0, q1, z) in SOC(p+2) then (beta+, beta-) >= 0:
        q0 - q1 = 1,   z - X beta+ + X beta- = -y,   minimise 1/2 (q0 + q1) + lam 1'(beta+ + beta-)
    (q0^2 - q1^2 = q0 + q1 >= ||z||^2).  Returns (data dict, cone dict): n = p + 2 + 2d, m = p + 1."""
    X, y, lam = lasso_data(p, d, density, seed)
    r1 = sp.hstack([sp.csc_matrix(np.array([[1.0, -1.0]])), sp.csc_matrix((1, p + 2 * d))])
    r2 = sp.hstack([sp.csc_matrix((p, 2)), sp.identity(p), -X, X])
    A = canonical_csc(sp.vstack([r1, r2]))
    b = np.concatenate([[1.0], -y])
    c = np.concatenate([[0.5, 0.5], np.zeros(p), lam * np.ones(2 * d)])
    return dict(A=A, b=b, c=c), dict(q=[p + 2], l=2 * d)
