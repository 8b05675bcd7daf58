"""GPU (-m gpu): the conic (ABIP-QCP) path through abip_qcp() against the CPU oracle (itself pinned on the reference's
recorded toy-QCP output) on the same inputs: iteration counts, (x, y, s), objectives; and the reference's recorded numbers
directly."""
import numpy as np
import pytest
import scipy.sparse as sp

from _golden import info_of, load, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
    from abip_amd import qcp
    return qcp


@pytest.fixture(scope="module")
def pq():
    from oracle import pyoracle_qcp
    pyoracle_qcp.lib()
    return pyoracle_qcp


def eps_all(eps):
    return dict(eps=eps, linsys_solver=1, verbose=0)


def toy():
    A = sp.csc_matrix(np.array([[1, 2, 3, 4, 5, 6, 7, 8], [0, 1, 2, 1, 2, 3, 1, 2]], dtype=float))
    return dict(A=A, b=np.array([4.0, 3.0]), c=np.array([1, 0, 2, 1, 4, 2, 3, 0], dtype=float), Q=sp.identity(8, format="csc")), dict(q=[3], rq=[3], f=1, l=1)


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


def test_toy_qcp_matches_recorded_reference_output(gpu):
    data, K = toy()
    sol, info = gpu.abip_qcp(data, K, eps_all(1e-6))
    assert info["status"] == "Solved" and info["ipm_iter"] == 10 and info["admm_iter"] == 91
    assert abs(info["pobj"] - (-0.984063813)) < 5e-9 and abs(info["dobj"] - (-0.984063938)) < 5e-9
    want = np.array([0.046341, 0.044938, 0.011319, 0.342543, 0.061490, 0.205246, -2.161307, 2.006235])
    assert np.max(np.abs(sol["x"] - want)) < 6e-7
    # the reference's per-phase timers (abip.c:1084-1093, 1196-1201): five non-negative totals that fit inside the solve time
    ph = info["phase_times"]
    assert set(ph) == {"project_lin_sys", "solve_barrier_subproblem", "calc_residuals", "err_inner", "updating_work"}
    assert all(v >= 0.0 for v in ph.values()) and ph["project_lin_sys"] > 0.0 and ph["solve_barrier_subproblem"] > 0.0
    # (the device phases are SAMPLED -- one bracketed iteration per control read, scaled to the iteration count -- so on a 91-iteration toy their
    #  sum may exceed the wall-clock solve time: no upper bound is asserted, only finiteness)
    assert all(np.isfinite(v) for v in ph.values())


@pytest.mark.parametrize("case", ["toy", "lasso_small", "lasso_mid", "lp_afiro", "lp_rand", "qp", "rsoc_mix", "lasso_bigcone"])
def test_conic_path_follows_the_oracle(gpu, pq, case):
    rng = np.random.default_rng(11)
    Q = None
    if case == "toy":
        data, K = toy(); Q = data["Q"]
    elif case == "lasso_small":
        data, K = lasso_socp(30, 60, 2)
    elif case == "lasso_mid":
        data, K = lasso_socp(400, 1500, 3, density=0.02)
    elif case == "lasso_bigcone":   # a cone above QC_BIG = 2048 entries takes the one-workgroup-per-cone kernel
        data, K = lasso_socp(2200, 2600, 4, density=0.004)
    elif case in ("lp_afiro", "lp_rand"):
        z, A, b, c = load("lp_afiro_like" if case == "lp_afiro" else "lp_random_sparse_small")
        data, K = dict(A=A, b=b, c=c), dict(l=A.shape[1])
    elif case == "qp":
        m2, n2 = 8, 20
        A2 = sp.random(m2, n2, density=0.4, random_state=rng, format="csc") + sp.hstack([sp.identity(m2), sp.csc_matrix((m2, n2 - m2))])
        G = rng.standard_normal((n2, n2)); Q = sp.csc_matrix(G @ G.T / n2 + 0.1 * np.eye(n2))
        data, K = dict(A=sp.csc_matrix(A2), b=A2 @ rng.random(n2), c=rng.standard_normal(n2), Q=Q), dict(l=n2)
    else:  # several SOC and rotated cones of different sizes + free + zero + orthant
        sizes_q, sizes_rq, f, zc, l, m2, dens = [3, 5, 1, 8], [3, 4, 6], 4, 2, 12, 9, 0.35
        n2 = sum(sizes_q) + sum(sizes_rq) + f + zc + l
        A2 = sp.random(m2, n2, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
        x0 = np.zeros(n2); pos = 0
        for sz in sizes_q:
            v = rng.standard_normal(sz); v[0] = np.linalg.norm(v[1:]) + 1.0; x0[pos:pos + sz] = v; pos += sz
        for sz in sizes_rq:
            v = rng.standard_normal(sz); v[0] = 1.0 + abs(v[0]); v[1] = (v[2:] @ v[2:]) / (2 * v[0]) + 0.5; x0[pos:pos + sz] = v; pos += sz
        x0[pos:pos + f] = rng.standard_normal(f); pos += f + zc
        x0[pos:] = rng.random(l) + 0.1
        data = dict(A=A2, b=A2 @ x0, c=A2.T @ rng.standard_normal(m2) + np.concatenate([x0[:sum(sizes_q) + sum(sizes_rq)], np.zeros(f), rng.standard_normal(zc), rng.random(l) + 0.1]))
        K = dict(q=sizes_q, rq=sizes_rq, f=f, z=zc, l=l)
    eps = {"lp_rand": 1e-4, "lasso_bigcone": 1e-3}.get(case, 1e-6)      # (the LP through the conic path needs ~1e5 iterations at 1e-6)
    x, y, s, oi, _ = pq.solve(data["A"], data["b"], data["c"], K, Q=Q, eps=eps, eps_p=eps, eps_d=eps, eps_g=eps, eps_inf=eps, eps_unb=eps, linsys_solver=1)
    sol, gi = gpu.abip_qcp(data, K, eps_all(eps))
    assert gi["status"] == oi["status"], (gi["status"], oi["status"])
    assert gi["ipm_iter"] == oi["ipm_iter"]
    assert abs(gi["admm_iter"] - oi["admm_iter"]) <= 0.03 * oi["admm_iter"] + 2
    if oi["status_val"] in (1, 2):
        tol = 1e-6 if gi["admm_iter"] == oi["admm_iter"] else 10 * eps
        assert rel(sol["x"], x) < tol and rel(sol["y"], y) < tol and rel(sol["s"], s) < max(tol, 1e-5)
        assert abs(gi["pobj"] - oi["pobj"]) <= tol * (1 + abs(oi["pobj"])) and abs(gi["dobj"] - oi["dobj"]) <= tol * (1 + abs(oi["dobj"]))
    if case in ("lp_afiro", "lp_rand"):
        # a pin that does not pass through the (unpinned) conic oracle: the same LP solved by the REAL LP reference (fixtures from
        # oracle/_ref at eps 1e-8) -- a different algorithm on the same problem, so agreement is at the conic run's own tolerance
        # (cf. the cross-solver check of the reference's test/test_abip_install.m:24-27)
        g = info_of(z, "direct_1e-08")
        lp_tol = 30 * eps
        assert abs(gi["pobj"] - g["pobj"]) <= lp_tol * (1 + abs(g["pobj"])) and abs(gi["dobj"] - g["dobj"]) <= lp_tol * (1 + abs(g["pobj"]))
        if case == "lp_rand":      # (the afiro-like LP has a face of optimal solutions: its objective and feasibility pin it, not x)
            assert rel(sol["x"], z["direct_1e-08_x"]) < 5e-2
        assert abs(c @ sol["x"] - g["pobj"]) <= lp_tol * (1 + abs(g["pobj"]))
        assert np.linalg.norm(A @ sol["x"] - b) / (1 + np.linalg.norm(b)) < 10 * eps and sol["x"].min() > -10 * eps


def test_conic_api_dispatch(gpu):
    import abip_amd
    data, K = toy()
    p = abip_amd.abip_get_params(); p.update(verbose=0, tol=1e-6)
    x, y, s, info = abip_amd.abip(data, K, p)          # K has q / rq / f  ->  abip_qcpsolve (abip.m:22-24)
    # abip_qcpsolve passes eps_p/d/g = tol but leaves eps_inf / eps_unb at 1e-3 (abip_qcpsolve.m:41-44); density > 0.4 would pick
    # linsys_solver 5 upstream -- the wrapper forces the direct solver 1 (pcg = 0)
    assert info["solver"] == "abip-qcp" and info["status"] == "Solved" and abs(info["pobj"] + 0.9840638) < 1e-5


def test_conic_tail_residual_guard(gpu, monkeypatch):
    """The set-up guard of the conic KKT factor: with the check forced to fail (the fault-injection hook lives only in libabip_hip_hooks.so, so the
    forced run happens in a child interpreter bound to that variant) the dense tail is dropped and the solve still follows."""
    import os, subprocess, sys, textwrap, json
    data, K = lasso_socp(400, 1500, 3, density=0.02)
    sol0, i0 = gpu.abip_qcp(data, K, eps_all(1e-5))
    assert i0["factor"]["dense_tail"] > 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, ABIP_HIP_TAIL_RESID_FAIL="1", ABIP_HIP_LIBRARY=os.path.join(root, "abip_amd", "lib", "libabip_hip_hooks.so"))
    code = textwrap.dedent(f"""
        import sys, json
        sys.path[:0] = [{root!r}, {os.path.join(root, 'tests')!r}]
        import numpy as np
        from abip_amd import qcp as gpu
        from test_gpu_qcp import eps_all, lasso_socp
        data, K = lasso_socp(400, 1500, 3, density=0.02)
        sol1, i1 = gpu.abip_qcp(data, K, eps_all(1e-5))
        print("RESULT " + json.dumps(dict(tail=i1["factor"]["dense_tail"], status=i1["status"], ipm=i1["ipm_iter"], pobj=i1["pobj"], x=sol1["x"].tolist())))
        """)
    r = subprocess.run([sys.executable, "-c", code], env=e, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.split("\n") if l.startswith("RESULT ")][-1][7:])
    assert res["tail"] == 0 and res["status"] == i0["status"] == "Solved" and res["ipm"] == i0["ipm_iter"]
    assert rel(np.array(res["x"]), sol0["x"]) < 1e-4 and abs(res["pobj"] - i0["pobj"]) < 1e-5 * (1 + abs(i0["pobj"]))


def test_device_transpose_that_fails_falls_back_to_the_host_transpose(gpu, monkeypatch):
    """The one library call on the device (hipcub's radix sort in dev_transpose.hip; everything else is hand-written) may refuse -- no room for its scratch, an
    error from the sort: DevCsr::from_columns then keeps nothing and the set-up transposes on the host as it does for small operators.  Forced through the hook of
    libabip_hip_hooks.so with the device transpose switched on for a small operator: same solve, bit for bit."""
    import os, subprocess, sys, textwrap, json
    data, K = lasso_socp(400, 1500, 3, density=0.02)
    monkeypatch.setenv("ABIP_HIP_DEV_TRANSPOSE", "1")
    sol0, i0 = gpu.abip_qcp(data, K, dict(eps=1e-5, linsys_solver=3, verbose=0))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, ABIP_HIP_DEV_TRANSPOSE="1", ABIP_HIP_DEV_TRANSPOSE_FAIL="1", ABIP_HIP_LIBRARY=os.path.join(root, "abip_amd", "lib", "libabip_hip_hooks.so"))
    code = textwrap.dedent(f"""
        import sys, json
        sys.path[:0] = [{root!r}, {os.path.join(root, 'tests')!r}]
        import numpy as np
        from abip_amd import qcp as gpu
        from test_gpu_qcp import lasso_socp
        data, K = lasso_socp(400, 1500, 3, density=0.02)
        sol1, i1 = gpu.abip_qcp(data, K, dict(eps=1e-5, linsys_solver=3, verbose=0))
        print("RESULT " + json.dumps(dict(status=i1["status"], ipm=i1["ipm_iter"], admm=int(i1["admm_iter"]), x=sol1["x"].tolist())))
        """)
    r = subprocess.run([sys.executable, "-c", code], env=e, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.split("\n") if l.startswith("RESULT ")][-1][7:])
    assert res["status"] == i0["status"] == "Solved" and res["ipm"] == i0["ipm_iter"] and res["admm"] == i0["admm_iter"]
    assert np.array_equal(np.array(res["x"]), sol0["x"])


def test_unsupported_back_ends_are_rejected(gpu):
    data, K = toy()
    # the reference's other exact factorisations (0 MKL-DSS, 2 Cholesky, 4 PARDISO, 5 LAPACK: what its default rule picks for dense data) run the device LDL'
    ref, ri = gpu.abip_qcp(data, K, dict(eps=1e-6, linsys_solver=1, verbose=0))
    for ls in (0, 2, 4, 5):
        sol, info = gpu.abip_qcp(data, K, dict(eps=1e-6, linsys_solver=ls, verbose=0))
        assert info["status"] == "Solved" and info["admm_iter"] == ri["admm_iter"] and np.array_equal(sol["x"], ref["x"])
    sol, info = gpu.abip_qcp(data, K, dict(eps=1e-3, linsys_solver=6, verbose=0))
    assert info["status"] == "Failure" and info["status_val"] == -4
    # the PCG back-end needs H = rho_x I + Q diagonal
    rng = np.random.default_rng(1)
    G = rng.standard_normal((8, 8)); data["Q"] = sp.csc_matrix(G @ G.T)
    sol, info = gpu.abip_qcp(data, K, dict(eps=1e-3, linsys_solver=3, verbose=0))
    assert info["status"] == "Failure" and info["status_val"] == -4


@pytest.mark.parametrize("case", ["lasso_small", "lasso_mid", "lasso_bigcone", "lasso_long_rows", "lp_afiro"])
def test_conic_pcg_with_the_gathered_vector_in_lds(gpu, case, monkeypatch):
    """qcp_pcg.h: kq_pcg_Aty_lds -- A' y with the m-vector resident in LDS (default from 2e6 non-zeros on where m <= 16 384; forced here): short rows (16 lanes per
    row), long rows (64), rows shorter than a lane group, an m-vector above 64 KB of LDS is covered by the full-size run of bench.py's c5 / lasso workloads.
    Same ADMM run as with the streaming kernel: status, outer iterations, inner iterations within 1 %, solution to 1e-6 relative of the objective scale."""
    if case == "lp_afiro":
        z, A, b, c = load("lp_afiro_like")
        data, K = dict(A=A, b=b, c=c), dict(l=A.shape[1])
    elif case == "lasso_long_rows":
        data, K = lasso_socp(300, 500, 6, density=0.5)      # ~150 non-zeros per column of X: the 64-lane form
    else:
        data, K = {"lasso_small": lambda: lasso_socp(30, 60, 2), "lasso_mid": lambda: lasso_socp(400, 1500, 3, density=0.02),
                   "lasso_bigcone": lambda: lasso_socp(2200, 2600, 4, density=0.004)}[case]()
    eps = 1e-3 if case == "lasso_bigcone" else 1e-5
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ABIP_HIP_ATY_LDS", mode)
        out[mode] = gpu.abip_qcp(data, K, dict(eps=eps, linsys_solver=3, verbose=0))
    (s0, i0), (s1, i1) = out["0"], out["1"]
    assert i0["status"] == i1["status"] and i0["status_val"] in (1, 2) and i0["ipm_iter"] == i1["ipm_iter"]
    assert abs(i0["admm_iter"] - i1["admm_iter"]) <= 0.01 * i0["admm_iter"] + 2
    assert abs(i0["pobj"] - i1["pobj"]) <= 1e-6 * (1 + abs(i0["pobj"]))
    for k in "xys":
        assert np.linalg.norm(s0[k] - s1[k]) <= 50 * eps * (1 + np.linalg.norm(s0[k]))


@pytest.mark.parametrize("case", ["toy", "lasso_small", "lasso_mid", "lp_afiro", "rsoc_mix", "lasso_bigcone"])
def test_conic_pcg_back_end(gpu, pq, case):
    """linsys_solver = 3: the device's y-space PCG (abip_amd/csrc/qcp_pcg.h; upstream's own conic PCG is unreachable and ill-posed, so the
    definition is this repository's, restated on the CPU by oracle/abip_qcp_oracle.c).  Checked three ways: against the oracle's PCG run
    (same algorithm, other summation order), against the device's direct back-end (another linear solver inside the same ADMM) and,
    for the LP case, against the real LP reference's fixture."""
    rng = np.random.default_rng(11)
    Q = None
    if case == "toy":
        data, K = toy(); Q = data["Q"]          # Q = I: diagonal
    elif case == "lasso_small":
        data, K = lasso_socp(30, 60, 2)
    elif case == "lasso_mid":
        data, K = lasso_socp(400, 1500, 3, density=0.02)
    elif case == "lasso_bigcone":
        data, K = lasso_socp(2200, 2600, 4, density=0.004)
    elif case == "lp_afiro":
        z, A, b, c = load("lp_afiro_like")
        data, K = dict(A=A, b=b, c=c), dict(l=A.shape[1])
    else:
        sizes_q, sizes_rq, f, zc, l, m2, dens = [3, 5, 1, 8], [3, 4, 6], 4, 2, 12, 9, 0.35
        n2 = sum(sizes_q) + sum(sizes_rq) + f + zc + l
        A2 = sp.random(m2, n2, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
        x0 = np.zeros(n2); pos = 0
        for sz in sizes_q:
            v = rng.standard_normal(sz); v[0] = np.linalg.norm(v[1:]) + 1.0; x0[pos:pos + sz] = v; pos += sz
        for sz in sizes_rq:
            v = rng.standard_normal(sz); v[0] = 1.0 + abs(v[0]); v[1] = (v[2:] @ v[2:]) / (2 * v[0]) + 0.5; x0[pos:pos + sz] = v; pos += sz
        x0[pos:pos + f] = rng.standard_normal(f); pos += f + zc
        x0[pos:] = rng.random(l) + 0.1
        data = dict(A=A2, b=A2 @ x0, c=A2.T @ rng.standard_normal(m2) + np.concatenate([x0[:sum(sizes_q) + sum(sizes_rq)], np.zeros(f), rng.standard_normal(zc), rng.random(l) + 0.1]))
        K = dict(q=sizes_q, rq=sizes_rq, f=f, z=zc, l=l)
    eps = {"lasso_bigcone": 1e-3, "lasso_mid": 1e-5}.get(case, 1e-6)
    st3 = dict(eps=eps, linsys_solver=3, verbose=0)
    x, y, s, oi, _ = pq.solve(data["A"], data["b"], data["c"], K, Q=Q, eps=eps, eps_p=eps, eps_d=eps, eps_g=eps, eps_inf=eps, eps_unb=eps, linsys_solver=3)
    sol, gi = gpu.abip_qcp(data, K, st3)
    # inexact inner solves: the two runs may leave the last outer iteration on different sides of the convergence test
    assert gi["status"] == oi["status"] and gi["status_val"] in (1, 2) and abs(gi["ipm_iter"] - oi["ipm_iter"]) <= 1
    same_outer = gi["ipm_iter"] == oi["ipm_iter"]
    if same_outer:
        assert abs(gi["admm_iter"] - oi["admm_iter"]) <= 0.03 * oi["admm_iter"] + 3
    tol = (20 if same_outer else 200) * eps
    assert rel(sol["x"], x) < tol and rel(sol["y"], y) < tol
    assert abs(gi["pobj"] - oi["pobj"]) <= tol * (1 + abs(oi["pobj"]))
    assert gi["avg_cg_iters"] > 0 and abs(gi["avg_cg_iters"] - oi["avg_cg_iters"]) <= 0.25 * oi["avg_cg_iters"] + 0.5
    sold, gd = gpu.abip_qcp(data, K, eps_all(eps))          # the direct back-end on the device
    assert gd["status"] == gi["status"] and abs(gd["ipm_iter"] - gi["ipm_iter"]) <= 1
    tol_d = 200 * eps
    assert abs(gd["pobj"] - gi["pobj"]) <= tol_d * (1 + abs(gd["pobj"])) and rel(sol["x"], sold["x"]) < max(tol_d, 5e-2 if case == "lp_afiro" else 0)
    if case == "lp_afiro":
        g = info_of(z, "direct_1e-08")
        assert abs(gi["pobj"] - g["pobj"]) <= 30 * eps * (1 + abs(g["pobj"]))



@pytest.mark.parametrize("p,d", [(4000, 18000), (10000, 45000)])
def test_lasso_at_config5_scale_properties(gpu, p, d):
    """BASELINE configs[4] (LASSO-as-SOCP, ~1e5 variables) at a reduced size and at the FULL size p = 10000, d = 45000 (n = 100 002).
    Too large for the CPU oracle in test time, so: size-independent properties.  The solution must
    satisfy the cone constraints and the reference's own residual criteria, and its LASSO objective must agree with the conic
    objective and beat the two trivial points beta = 0 and a proximal-gradient iterate."""
    from abip_amd import problems
    data, K = problems.qcp_lasso_socp(p, d)
    eps = 1e-4
    sol, info = gpu.abip_qcp(data, K, eps_all(eps))
    assert info["status"] == "Solved" and info["res_pri"] < eps and info["res_dual"] < eps and info["gap"] < eps
    x = sol["x"]
    q0, q1, z = x[0], x[1], x[2:p + 2]
    bp, bm = x[p + 2:p + 2 + d], x[p + 2 + d:]
    assert q0 >= np.sqrt(q1 * q1 + z @ z) * (1 - 1e-6) and bp.min() > -1e-9 and bm.min() > -1e-9      # x in K
    A, b, c = data["A"], data["b"], data["c"]
    assert np.linalg.norm(A @ x - b) <= 1e-3 * (1 + np.linalg.norm(b))
    X = -A[1:, p + 2:p + 2 + d]; yv = -b[1:]; lam = c[-1]
    beta = bp - bm
    lasso = lambda bb: 0.5 * np.sum((X @ bb - yv) ** 2) + lam * np.abs(bb).sum()
    assert abs(lasso(beta) - info["pobj"]) <= 5e-3 * (1 + abs(info["pobj"]))
    assert lasso(beta) < lasso(np.zeros(d))
    # KKT of the LASSO itself: |X'(X beta - y)|_inf <= lam (1 + small)
    g = X.T @ (X @ beta - yv)
    assert np.abs(g).max() <= lam * (1 + 5e-2)


def test_cone_kernel_matches_the_oracle_on_every_branch(gpu, pq):
    """kq_cones against the oracle's restatement of cones.c:130-248, cone by cone: |a| <= 1e-9, a > 0, a < 0 for the SOC; ze + zn
    == 0 (reads the incoming x[0]), > 0, < 0 with w <= 10 and w > 10 for the rotated cone; lengths 1..2500 so that the
    one-wavefront and the one-workgroup (> 2048 entries) kernels both run.  The device reduces |tail|^2 in a different order."""
    import importlib
    tq = importlib.import_module("test_qcp_oracle")
    from abip_amd import qcp
    rng = np.random.default_rng(0)
    n_big = 0
    for kind, t, xp, lam in tq._soc_cases(rng):
        want = pq.cone_prox(kind, t, lam, xp)
        got = qcp.cone_prox(kind, t, lam, xp)
        assert np.all(np.isfinite(got))
        assert np.max(np.abs(got - want)) <= 1e-13 * (1 + np.max(np.abs(want))), (kind, t.size, t[:2], lam)
        n_big += t.size > 2048
    assert n_big >= 8


@pytest.mark.parametrize("case", ["toy", "lasso_mid"])
def test_conic_batched_iterations_equal_stepwise(gpu, case, monkeypatch):
    """The inner exit test is evaluated by kq_finalize and iterations are enqueued in batches between residual checks; one control
    read per iteration (ABIP_HIP_BATCH=0) must give the same run bit for bit."""
    data, K = toy() if case == "toy" else lasso_socp(400, 1500, 3, density=0.02)
    runs = []
    for mode in ("batched", "stepwise", "stepwise", "batched", "stepwise"):     # (repeats: the runs must also be reproducible)
        if mode == "stepwise":
            monkeypatch.setenv("ABIP_HIP_BATCH", "0")
        else:
            monkeypatch.delenv("ABIP_HIP_BATCH", raising=False)
        sol, info = gpu.abip_qcp(data, K, eps_all(1e-6))
        runs.append((info["admm_iter"], info["ipm_iter"], info["pobj"], sol["x"].copy(), sol["y"].copy(), sol["s"].copy()))
    for other in runs[1:]:
        assert runs[0][:3] == other[:3]
        for a2, b2 in zip(runs[0][3:], other[3:]):
            assert np.array_equal(a2, b2)


@pytest.mark.parametrize("shape", [(1, 1, 1.0), (7, 5, 0.6), (300, 900, 0.05), (5001, 3000, 0.01), (64, 70000, 0.002), (70000, 64, 0.002)])
def test_device_transpose_equals_the_host(gpu, shape):
    """dev_transpose.hip: the row form of an operator built on the device from its column form (a stable sort of the entry numbers by row) is, entry for entry,
    what the host's counting sort (host_par.h; reference: indirect.c:81-139) leaves: row pointers, columns ascending inside a row, the values bit for bit.
    Empty rows and columns, one row, one column, more rows than columns and the other way round."""
    import ctypes as C
    from abip_amd import _lib
    L = _lib.load()
    m, n, dens = shape
    rng = np.random.default_rng(m * 31 + n)
    A = sp.random(m, n, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    if A.nnz == 0:
        A = sp.csc_matrix(np.ones((m, n)))
    A.sort_indices()
    Ap, Ai, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    ptr, col, val = np.zeros(m + 1, np.int32), np.zeros(A.nnz, np.int32), np.zeros(A.nnz)
    PI32 = C.POINTER(C.c_int)
    L.abip_hip_csc_to_csr.restype = C.c_int
    L.abip_hip_csc_to_csr.argtypes = [C.c_int, C.c_int, PI32, PI32, _lib.PF, PI32, PI32, _lib.PF]
    rc = L.abip_hip_csc_to_csr(m, n, Ap.ctypes.data_as(PI32), Ai.ctypes.data_as(PI32), Ax.ctypes.data_as(_lib.PF), ptr.ctypes.data_as(PI32), col.ctypes.data_as(PI32), val.ctypes.data_as(_lib.PF))
    assert rc == 0, rc
    R = A.tocsr(); R.sort_indices()
    assert np.array_equal(ptr, R.indptr) and np.array_equal(col, R.indices) and np.array_equal(val, R.data)


@pytest.mark.parametrize("ls", [1, 3])
def test_conic_solve_is_the_same_with_the_device_transpose(gpu, ls, monkeypatch):
    """The conic set-up transposes large operators on the device (from 2e6 non-zeros; forced here on a small one): the same arrays reach the kernels, so the solve is
    bit-identical to the one over the host's row form."""
    data, K = lasso_socp(400, 1500, 3, density=0.02)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ABIP_HIP_DEV_TRANSPOSE", mode)
        out[mode] = gpu.abip_qcp(data, K, dict(eps=1e-5, linsys_solver=ls, verbose=0))
    (s0, i0), (s1, i1) = out["0"], out["1"]
    assert i0["status_val"] == 1 and i0["admm_iter"] == i1["admm_iter"] and i0["ipm_iter"] == i1["ipm_iter"]
    for k in "xys":
        assert np.array_equal(s0[k], s1[k])
