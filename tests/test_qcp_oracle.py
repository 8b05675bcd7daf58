"""CPU: pin the QCP oracle (oracle/abip_qcp_oracle.c).  There is no oracle/_ref for QCP (its sources need MKL
headers, see the oracle's header), so the pins are: the reference's own recorded output on the literal toy problem of
test/test_abip_install.m:32-43, the cross-solver LP check of test_abip_install.m:24-27, and
optimality properties."""
import numpy as np
import pytest
import scipy.sparse as sp

from _golden import info_of, load, rel


@pytest.fixture(scope="module")
def pq():
    from oracle import pyoracle_qcp
    pyoracle_qcp.lib()
    return pyoracle_qcp


def toy():
    A = sp.csc_matrix(np.array([[1, 2, 3, 4, 5, 6, 7, 8], [0, 1, 2, 1, 2, 3, 1, 2]], dtype=float))
    return A, np.array([4.0, 3.0]), np.array([1, 0, 2, 1, 4, 2, 3, 0], dtype=float), sp.identity(8, format="csc"), dict(q=[3], rq=[3], f=1, l=1)


def eps_all(eps):
    return dict(eps=eps, eps_p=eps, eps_d=eps, eps_g=eps, eps_inf=eps, eps_unb=eps, linsys_solver=1)


def test_toy_qcp_matches_recorded_reference_output(pq):
    """reference, linsys_solver=1, eps=1e-6 -> Solved ipm=10 admm=91 pobj=-0.984063813 dobj=-0.984063938."""
    A, b, c, Q, K = toy()
    x, y, s, info, _ = pq.solve(A, b, c, K, Q=Q, **eps_all(1e-6))
    assert info["status"] == "Solved" and info["ipm_iter"] == 10 and info["admm_iter"] == 91
    assert abs(info["pobj"] - (-0.984063813)) < 5e-10 and abs(info["dobj"] - (-0.984063938)) < 5e-10
    want = np.array([0.046341, 0.044938, 0.011319, 0.342543, 0.061490, 0.205246, -2.161307, 2.006235])
    assert np.max(np.abs(x - want)) < 6e-7


@pytest.mark.parametrize("name", ["lp_afiro_like", "lp_random_sparse_small"])
def test_lp_through_the_conic_path_reaches_the_lp_optimum(pq, name):
    """test_abip_install.m:24-27 runs the same LP through the QCP solver (param.solver = 1)."""
    z, A, b, c = load(name)
    g = info_of(z, "direct_1e-06")
    x, y, s, info, _ = pq.solve(A, b, c, dict(l=A.shape[1]), **eps_all(1e-6))
    assert info["status"] == "Solved"
    assert abs(info["pobj"] - g["pobj"]) <= 2e-5 * (1 + abs(g["pobj"]))
    # (x itself is not compared: these LPs have non-unique optimal vertices and the two algorithms pick different ones)
    assert abs(info["pobj"] - info["dobj"]) <= 2e-5 * (1 + abs(info["pobj"]))
    assert np.linalg.norm(A @ x - b) / (1 + np.linalg.norm(b)) < 1e-5 and x.min() > -1e-6


def test_socp_and_qp_optimality_conditions(pq):
    """LASSO as SOCP (SURVEY.md 8(d), config 5 in miniature) and a small convex QP: KKT residuals of the returned point."""
    rng = np.random.default_rng(2)
    p, dft = 30, 60
    X = sp.random(p, dft, density=0.2, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    yv = X @ (rng.standard_normal(dft) * (rng.random(dft) < 0.3)) + 0.01 * rng.standard_normal(p)
    lam = np.abs(X.T @ yv).max() / 5
    # variables: q = (q0, q1, z[p]) in SOC(p+2), then beta+ (d), beta- (d) >= 0;  q0 - q1 = 1,  z - X b+ + X b- = -y
    n = p + 2 + 2 * dft
    r1 = sp.hstack([sp.csc_matrix(np.array([[1.0, -1.0]])), sp.csc_matrix((1, p + 2 * dft))])
    r2 = sp.hstack([sp.csc_matrix((p, 2)), sp.identity(p), -X, X])
    A = sp.vstack([r1, r2]).tocsc()
    b = np.concatenate([[1.0], -yv])
    c = np.concatenate([[0.5, 0.5], np.zeros(p), lam * np.ones(2 * dft)])
    x, y, s, info, _ = pq.solve(A, b, c, dict(q=[p + 2], l=2 * dft), **eps_all(1e-7))
    assert info["status"] == "Solved"
    beta = x[p + 2: p + 2 + dft] - x[p + 2 + dft:]
    obj = 0.5 * np.sum((X @ beta - yv) ** 2) + lam * np.abs(beta).sum()
    assert abs(info["pobj"] - obj) <= 1e-4 * (1 + abs(obj))
    # first-order optimality of LASSO: |X'(X beta - y)|_inf <= lam (+ tolerance)
    assert np.abs(X.T @ (X @ beta - yv)).max() <= lam * (1 + 1e-3)
    assert np.linalg.norm(A @ x - b) < 1e-5 * (1 + np.linalg.norm(b))
    assert x[0] >= np.linalg.norm(x[1: p + 2]) - 1e-6 and s[0] >= np.linalg.norm(s[1: p + 2]) - 1e-6     # primal / dual cone
    assert abs(x @ s) <= 1e-4 * (1 + abs(info["pobj"]))                                                    # complementarity
    # QP: min 1/2 x'Qx + c'x, Ax = b, x >= 0
    m2, n2 = 8, 20
    A2 = sp.random(m2, n2, density=0.4, random_state=rng, format="csc") + sp.hstack([sp.identity(m2), sp.csc_matrix((m2, n2 - m2))])
    G = rng.standard_normal((n2, n2)); Q2 = sp.csc_matrix(G @ G.T / n2 + 0.1 * np.eye(n2))
    b2 = A2 @ rng.random(n2); c2 = rng.standard_normal(n2)
    x2, y2, s2, info2, _ = pq.solve(A2, b2, c2, dict(l=n2), Q=Q2, **eps_all(1e-7))
    assert info2["status"] == "Solved"
    assert np.linalg.norm(A2 @ x2 - b2) < 1e-5 * (1 + np.linalg.norm(b2))
    assert np.linalg.norm(Q2 @ x2 + c2 - A2.T @ y2 - s2) < 1e-4 * (1 + np.linalg.norm(c2))
    assert x2.min() > -1e-6 and s2.min() > -1e-6 and abs(x2 @ s2) < 1e-4


def _soc_cases(rng):
    """(kind, tmp, x_prev, lambda) covering every branch of cones.c:130-248."""
    out = []
    for n in (1, 2, 3, 70, 2500):
        for a in (1.5, -0.7, 1e-10, 0.0):                                   # a > 0, a < 0, |a| <= 1e-9 (cones.c:137)
            out.append((0, np.concatenate([[a], rng.standard_normal(n - 1)]), None, 0.3))
    for n in (2, 3, 70, 2500):
        zx = rng.standard_normal(n - 2)
        out.append((1, np.concatenate([[0.8, 0.5], zx]), None, 0.3))        # ze + zn > 0
        out.append((1, np.concatenate([[-0.8, -0.5], 0.1 * zx]), None, 2.0))  # ze + zn < 0, w <= 10
        out.append((1, np.concatenate([[-30.0, -20.0], 0.01 * zx]), None, 1e-3))  # ze + zn < 0, w > 10 (cones.c:228)
        out.append((1, np.concatenate([[0.6, -0.6], zx]), rng.standard_normal(n), 0.3))  # ze + zn == 0 (cones.c:177-186)
        out.append((1, np.concatenate([[2.0, 1.0], 0.1 * zx]), None, 0.5))  # 2 ze zn - |zx|^2 >= 0 when n small
    return out


def test_cone_prox_is_the_minimiser_of_the_barrier_subproblem(pq):
    """Independent of the reference's code: x = prox must be interior and satisfy x - t = lambda * grad log det(x), the
    stationarity condition of  min -lambda log(det x) + 1/2 |x - t|^2  with det = x0^2 - |x1|^2 (SOC) or 2 x0 x1 - |x2|^2
    (rotated).  The `ze + zn == 0` shortcut of the reference is not a minimiser (it reuses the previous x[0], cones.c:183) and the
    |a| <= 1e-9 branch is the a -> 0 limit; both are excluded here and pinned through full solves instead."""
    rng = np.random.default_rng(0)
    seen = set()
    for kind, t, xp, lam in _soc_cases(rng):
        if (kind == 0 and abs(t[0]) <= 1e-9) or (kind == 1 and t[0] + t[1] == 0):
            continue
        x = pq.cone_prox(kind, t, lam, xp)
        if kind == 0:
            det = x[0] ** 2 - x[1:] @ x[1:]
            g = np.concatenate([[2 * x[0]], -2 * x[1:]]) / det
            assert x[0] > 0
        else:
            det = 2 * x[0] * x[1] - x[2:] @ x[2:]
            g = np.concatenate([[2 * x[1], 2 * x[0]], -2 * x[2:]]) / det
            assert x[0] > 0 and x[1] > 0
        assert det > 0
        assert np.linalg.norm(x - t - lam * g) <= 1e-9 * (1 + np.linalg.norm(t)), (kind, t[:2], lam)
        seen.add((kind, t.size))
    assert len(seen) >= 8
    t = np.array([3.0, -2.0, 1e-3, -50.0])                                   # orthant: x - t = lambda / x
    x = pq.cone_prox(2, t, 0.25)
    assert np.all(x > 0) and np.allclose(x - t, 0.25 / x, rtol=1e-12)


@pytest.mark.parametrize("case", ["toy", "lasso"])
def test_oracle_pcg_back_end_agrees_with_its_direct_back_end(pq, case):
    """linsys_solver = 3 (y-space PCG on rho_y I + A H^-1 A', the definition of abip_amd/csrc/qcp_pcg.h) inside the same ADMM must land where the
    direct back-end lands: same status and outer count, inner count within a few iterations, solution to the run's tolerance."""
    import scipy.sparse as sp
    if case == "toy":
        A = sp.csc_matrix(np.array([[1, 2, 3, 4, 5, 6, 7, 8], [0, 1, 2, 1, 2, 3, 1, 2]], dtype=float))
        b, c, Q, K = np.array([4.0, 3.0]), np.array([1, 0, 2, 1, 4, 2, 3, 0], dtype=float), sp.identity(8, format="csc"), dict(q=[3], rq=[3], f=1, l=1)
    else:
        rng = np.random.default_rng(2)
        p_, dft = 30, 60
        X = sp.random(p_, dft, density=0.2, random_state=rng, data_rvs=rng.standard_normal, format="csc")
        yv = X @ (rng.standard_normal(dft) * (rng.random(dft) < 0.3)) + 0.01 * rng.standard_normal(p_)
        lam = np.abs(X.T @ yv).max() / 5
        A = sp.vstack([sp.hstack([sp.csc_matrix(np.array([[1.0, -1.0]])), sp.csc_matrix((1, p_ + 2 * dft))]), sp.hstack([sp.csc_matrix((p_, 2)), sp.identity(p_), -X, X])]).tocsc()
        b = np.concatenate([[1.0], -yv]); c = np.concatenate([[0.5, 0.5], np.zeros(p_), lam * np.ones(2 * dft)]); Q = None; K = dict(q=[p_ + 2], l=2 * dft)
    out = {}
    for ls in (1, 3):
        x, y, s, oi, _ = pq.solve(A, b, c, K, Q=Q, eps=1e-6, eps_p=1e-6, eps_d=1e-6, eps_g=1e-6, eps_inf=1e-6, eps_unb=1e-6, linsys_solver=ls)
        out[ls] = (x, oi)
    a, p3 = out[1], out[3]
    assert a[1]["status"] == p3[1]["status"] == "Solved" and a[1]["ipm_iter"] == p3[1]["ipm_iter"]
    assert abs(a[1]["admm_iter"] - p3[1]["admm_iter"]) <= 0.03 * a[1]["admm_iter"] + 3
    assert np.linalg.norm(a[0] - p3[0]) / np.linalg.norm(a[0]) < 1e-6 and abs(a[1]["pobj"] - p3[1]["pobj"]) < 1e-5 * (1 + abs(a[1]["pobj"]))
    assert p3[1]["avg_cg_iters"] > 0 and a[1]["avg_cg_iters"] == 0
