"""GPU (-m gpu): the HIP path through the C ABI against (i) the committed outputs of the real reference
(tests/golden), (ii) the CPU oracle on the same seeded inputs, (iii) size-independent properties at the full
BASELINE size.

Tolerances (fp64 everywhere; the device sums in a different order than the CPU):
  * SpMV / KKT-solve unit checks: 1e-13 relative (direct), CG tolerance for the PCG back-end;
  * per-iteration iterates vs the oracle: 1e-9 relative;
  * final (x, y, s): 10*eps against the reference's fixtures at the fixture's eps, and 1e-6 relative against the oracle
    when both run to eps = 1e-8 (see _check_against_golden for why a run at tolerance eps is defined only up to O(eps))."""
import numpy as np
import pytest
import scipy.sparse as sp

from _golden import TINY_VARIANTS, info_of, load, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    import abip_amd
    return abip_amd


def run_with_hooks(body: str, env: dict):
    """The fault-injection hooks (ABIP_HIP_*_FAIL) exist only in libabip_hip_hooks.so (built -DABIP_HIP_TEST_HOOKS); the shipped library ignores the
    variables.  A library is bound once per process, so the body runs in a child interpreter that loads the hooks variant (ABIP_HIP_LIBRARY)."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    e["ABIP_HIP_LIBRARY"] = os.path.join(root, "abip_amd", "lib", "libabip_hip_hooks.so")
    pre = textwrap.dedent(f"""
        import sys
        sys.path[:0] = [{root!r}, {os.path.join(root, 'tests')!r}]
        import numpy as np, scipy.sparse as sp
        import abip_amd as gpu
        from _golden import load, rel
        def kkt_matrix(Asc, rho):
            m, n = Asc.shape
            return sp.bmat([[rho * sp.identity(m), Asc], [Asc.T, -sp.identity(n)]], format="csc")
        """)
    r = subprocess.run([sys.executable, "-c", pre + textwrap.dedent(body)], env=e, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])


def kkt_matrix(Asc, rho):
    m, n = Asc.shape
    return sp.bmat([[rho * sp.identity(m), Asc], [Asc.T, -sp.identity(n)]], format="csc")


# ---------------------------------------------------------------------------------------------- unit kernels
@pytest.mark.parametrize("name", ["lp_afiro_like", "lp_staircase", "lp_multicommodity_small"])
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_spmv_and_kkt_solve(gpu, name, linsys):
    z, A, b, c = load(name)
    rng = np.random.default_rng(1)
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0) as S:
        Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
        x, y = rng.standard_normal(S.n), rng.standard_normal(S.m)
        assert rel(S.accum_by_A(x, y), y + Asc @ x) < 1e-14
        assert rel(S.accum_by_Atrans(y, x), x + Asc.T @ y) < 1e-14
        K = kkt_matrix(Asc, 1e-3)
        rhs = rng.standard_normal(S.m + S.n)
        sol, its = S.kkt_solve(rhs, None, -1)
        if linsys == "direct":
            assert its == 0 and rel(K @ sol, rhs) < 1e-11
        else:
            assert its > 0 and rel(K @ sol, rhs) < 1e-6          # cg_tol = 1e-9*||b_y|| floored at 1e-7 (indirect.c:406-409)
            sol2, its2 = S.kkt_solve(rhs, sol[: S.m], -1)         # warm start at the solution: nothing left to do
            assert its2 <= 2


@pytest.mark.parametrize("tail", ["0", "64", "128", "512", "auto"])
def test_direct_solve_dense_tail(gpu, oracle_built, tail, monkeypatch):
    """Head/tail split of the factor (dev_ldl.h): the same K^-1 rhs whatever part of L is applied as a dense inverse, and the
    same ADMM trajectory.  `0` = pure level-scheduled solve, `auto` = trailing block chosen by density."""
    if tail == "auto":
        monkeypatch.delenv("ABIP_HIP_TAIL", raising=False)
    else:
        monkeypatch.setenv("ABIP_HIP_TAIL", tail)
    z, A, b, c = load("lp_staircase")
    rng = np.random.default_rng(7)
    with gpu.Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=400) as S:
        T = int(S.scalar("tail"))
        assert (T >= 256 and S.scalar("levels_fwd") <= 16) if tail == "auto" else T == int(tail), T
        if T == 0:
            assert S.scalar("levels_fwd") > 100             # what the tail is there to remove
        Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
        K = kkt_matrix(Asc, 1e-3)
        for _ in range(3):
            rhs = rng.standard_normal(S.m + S.n)
            sol, its = S.kkt_solve(rhs, None, -1)
            assert its == 0 and rel(K @ sol, rhs) < 1e-11
        info = S.solve()
    ref = oracle_built.solve("oracle", A, b, c, linsys="direct", verbose=0, max_admm_iters=400)
    assert info["admm_iter"] == ref.info["admm_iter"] and info["ipm_iter"] == ref.info["ipm_iter"]
    assert abs(info["pobj"] - ref.info["pobj"]) <= 1e-8 * (1 + abs(ref.info["pobj"]))


@pytest.mark.parametrize("tail", ["64", "128", "512", "auto"])
def test_tail_as_one_symmetric_matvec(gpu, oracle_built, tail, monkeypatch):
    """dev_ldl.h: k_tail_sym -- M = W' D2^-1 W formed once, each solve streams its lower triangle once (the default from 2 048 tail pivots on; forced here on
    small tails, including ones that do not fill a 512-column tile).  Same K^-1 rhs as the two triangular mat-vecs to 1e-10, same ADMM run as the oracle."""
    if tail == "auto":
        monkeypatch.delenv("ABIP_HIP_TAIL", raising=False)
    else:
        monkeypatch.setenv("ABIP_HIP_TAIL", tail)
    z, A, b, c = load("lp_staircase")
    rng = np.random.default_rng(13)
    rhs = [rng.standard_normal(A.shape[0] + A.shape[1]) for _ in range(3)]
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ABIP_HIP_TAIL_SYM", mode)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=400) as S:
            assert int(S.scalar("tail")) > 0
            Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
            K = kkt_matrix(Asc, 1e-3)
            sols = [S.kkt_solve(r, None, -1)[0] for r in rhs]
            for r, x in zip(rhs, sols):
                assert rel(K @ x, r) < (1e-11 if mode == "0" else 2e-10)   # the explicit inverse of S costs a digit on this (degenerate) LP; the set-up guard sits at 1e-8
            out[mode] = (sols, S.solve())
    for a, b_ in zip(out["0"][0], out["1"][0]):
        assert rel(a, b_) < 1e-10
    ref = oracle_built.solve("oracle", A, b, c, linsys="direct", verbose=0, max_admm_iters=400)
    for mode in ("0", "1"):
        assert out[mode][1]["admm_iter"] == ref.info["admm_iter"] and out[mode][1]["ipm_iter"] == ref.info["ipm_iter"]
        assert abs(out[mode][1]["pobj"] - ref.info["pobj"]) <= 1e-8 * (1 + abs(ref.info["pobj"]))


def test_symmetric_tail_self_check_keeps_the_two_matvecs(gpu, monkeypatch):
    """Before W and W' are released the set-up compares M v with W' D2^-1 W v (dev_ldl.h); a difference above 1e-9 (forced by the hook) keeps the two triangular
    mat-vecs.  On this LP their residual is < 1e-11 where the explicit inverse of S gives ~6e-11: the bound below shows which form answered."""
    monkeypatch.delenv("ABIP_HIP_TAIL", raising=False)
    run_with_hooks("""
        z, A, b, c = load("lp_staircase")
        rng = np.random.default_rng(17)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=50) as S:
            assert int(S.scalar("tail")) >= 256
            Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
            K = kkt_matrix(Asc, 1e-3)
            for _ in range(3):
                rhs = rng.standard_normal(S.m + S.n)
                sol, _ = S.kkt_solve(rhs, None, -1)
                assert rel(K @ sol, rhs) < 1e-11
        """, {"ABIP_HIP_TAIL_SYM": "1", "ABIP_HIP_TAIL_SYM_FAIL": "1", "ABIP_HIP_XCD": "0"})


@pytest.mark.parametrize("tail", ["128", "512", "auto"])
def test_schur_complement_formed_on_the_device(gpu, tail, monkeypatch):
    """LdlHost::dev_schur: the host hands over K22 and L21, the device subtracts L21 D1 L21' with dense panels (dev_ldl.h: k_l21_panel,
    k_schur_sub; or row by row from the sparse L21, k_schur_rows) before factoring the tail.  Same K^-1 rhs and the same ADMM run as with the host's sparse accumulation."""
    if tail == "auto":
        monkeypatch.delenv("ABIP_HIP_TAIL", raising=False)
    else:
        monkeypatch.setenv("ABIP_HIP_TAIL", tail)
    z, A, b, c = load("lp_staircase")
    rng = np.random.default_rng(11)
    rhs = [rng.standard_normal(A.shape[0] + A.shape[1]) for _ in range(2)]
    out = {}
    for mode in ("0", "1", "2"):   # host; device, dense panels; device, row-wise sparse kernel
        monkeypatch.setenv("ABIP_HIP_DEV_SCHUR", mode)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=300) as S:
            assert int(S.scalar("tail")) > 0
            Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
            K = kkt_matrix(Asc, 1e-3)
            sols = [S.kkt_solve(r, None, -1)[0] for r in rhs]
            for r, x in zip(rhs, sols):
                assert rel(K @ x, r) < 1e-11
            out[mode] = (sols, S.solve())
    for mode in ("1", "2"):
        for a, b_ in zip(out["0"][0], out[mode][0]):
            assert rel(a, b_) < 1e-10
        assert out["0"][1]["admm_iter"] == out[mode][1]["admm_iter"] and abs(out["0"][1]["pobj"] - out[mode][1]["pobj"]) <= 1e-9 * (1 + abs(out["0"][1]["pobj"]))


@pytest.mark.parametrize("name", ["lp_afiro_like", "lp_staircase"])
def test_batched_iterations_equal_stepwise(gpu, name, monkeypatch):
    """Direct back-end: iterations are enqueued in batches and the device finds the inner-loop exit (k_finalize raises halt).
    That must be bit-identical to one control read per iteration, and to stepping through the public ABI in odd strides."""
    z, A, b, c = load(name)
    runs = []
    for mode in ("batched", "stepwise", "strided"):
        if mode == "stepwise":
            monkeypatch.setenv("ABIP_HIP_BATCH", "0")
        else:
            monkeypatch.delenv("ABIP_HIP_BATCH", raising=False)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0, eps=1e-5) as S:
            if mode == "strided":
                S.begin()
                fin, total = False, 0
                while not fin:
                    fin, done = S.step(7)
                    total += done
                    assert done <= 7
                info = S.end()
                assert total == info["admm_iter"] - 1          # get_info reports k + 1 (abip.c:1296-1340)
            else:
                info = S.solve()
            runs.append((info["admm_iter"], info["ipm_iter"], info["pobj"], S.x.copy(), S.y.copy(), S.s.copy()))
    for r in runs[1:]:
        assert r[0] == runs[0][0] and r[1] == runs[0][1] and r[2] == runs[0][2]
        for a2, b2 in zip(r[3:], runs[0][3:]):
            assert np.array_equal(a2, b2)


@pytest.mark.parametrize("variant", ["default", "origin", "qp", "scale5", "nonorm"])
def test_host_scaling_matches_the_oracle(gpu, oracle_built, variant):
    """a12: ABIP(_normalize_A) (linsys/common.c:150-565: pc / origin / Ruiz x10 / qp rescaling with their clamps),
    normalize_b_c, sc_b, sc_c and the scaled h = (-b, c): the device's copies against the oracle's after the same set-up."""
    z, A, b, c = load("lp_staircase")
    kw = dict(TINY_VARIANTS.get(variant, {}))
    o = oracle_built.solve("oracle", A, b, c, linsys="indirect", max_admm_iters=3, **kw)
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, max_admm_iters=3, **kw) as S:
        S.begin()
        for nm in ("b", "c", "h") if variant == "nonorm" else ("D", "E", "b", "c", "h"):   # no D, E without normalisation
            assert rel(S.vector(nm), o.work[nm]) < 1e-14, (variant, nm)
        for nm in () if variant == "nonorm" else ("sc_b", "sc_c"):
            assert abs(S.scalar(nm) - o.work[nm]) <= 1e-14 * abs(o.work[nm]), (variant, nm)
        Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
        if variant != "nonorm":     # A_scaled = scale * D^-1 A E^-1
            want = sp.diags(1.0 / o.work["D"]) @ A @ sp.diags(1.0 / o.work["E"]) * float(kw.get("scale", 1.0))
            assert abs(Asc - want).max() <= 1e-13 * abs(want).max()
        S.end()


def _dense_lp(m, n, density, seed):
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    A = sp.csc_matrix(A + sp.hstack([sp.identity(m), sp.csc_matrix((m, n - m))]))
    x0 = rng.random(n) + 0.1
    return A, A @ x0, rng.random(n) + 0.1


@pytest.mark.parametrize("linsys", ["indirect", "direct"])
@pytest.mark.parametrize("shape", ["1x2", "2x3", "dense_mid", "dense_full", "tall_warned"])
def test_edge_shapes_and_sparsity_branches(gpu, oracle_built, shape, linsys):
    """Tiny systems, and the two inner-loop caps that depend on the sparsity of A (abip.c:2104-2115: sp > 0.5 -> mu^-0.35 iterations,
    0.2 < sp <= 0.5 -> 1/mu iterations), which no other fixture reaches; `tall_warned` has m close to n (validation warns, solves)."""
    if shape == "1x2":
        A, b, c = sp.csc_matrix(np.array([[1.0, 2.0]])), np.array([2.0]), np.array([1.0, 1.0])
    elif shape == "2x3":
        A, b, c = sp.csc_matrix(np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 1.0]])), np.array([1.0, 1.5]), np.array([1.0, 0.5, 2.0])
    elif shape == "dense_mid":
        A, b, c = _dense_lp(12, 40, 0.3, 1)        # 0.2 < sp <= 0.5
    elif shape == "dense_full":
        A, b, c = _dense_lp(10, 30, 0.9, 2)        # sp > 0.5
    else:
        A, b, c = _dense_lp(30, 34, 0.2, 3)
    sp_ratio = A.nnz / (A.shape[0] * A.shape[1])
    if shape == "dense_mid":
        assert 0.2 < sp_ratio <= 0.5
    if shape == "dense_full":
        assert sp_ratio > 0.5
    cap = 20000 if shape == "tall_warned" else 100000   # (the PCG back-end runs `tall_warned` into its cap: no need for a long one)
    o = oracle_built.solve("oracle", A, b, c, linsys=linsys, eps=1e-6, max_admm_iters=cap)
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-6, max_admm_iters=cap) as S:
        info = S.solve()
        assert info["status_val"] == o.info["status_val"], (shape, info["status"], o.info["status"])
        if o.info["status_val"] == 1:       # (the PCG back-end stalls on `tall_warned` at its 1e-7 CG floor -- reference, oracle and device
            assert info["ipm_iter"] == o.info["ipm_iter"]   # all end "Solved/Inaccurate" at the iteration cap; nothing finer to compare there)
            assert info["admm_iter"] == o.info["admm_iter"], (shape, linsys, info["admm_iter"], o.info["admm_iter"])
            assert abs(info["pobj"] - o.info["pobj"]) <= 1e-5 * (1 + abs(o.info["pobj"]))
            for k in "xys":
                assert rel(getattr(S, k), getattr(o, k)) < 1e-5, (shape, linsys, k)


@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_restart_path_follows_the_oracle(gpu, oracle_built, linsys):
    """restart_vars (abip.c:587-630) with the threshold lowered so that the periodic restart from the running mean fires many
    times; on the direct back-end a batch of iterations must stop short of an iteration that restarts."""
    from abip_amd import problems
    A, b, c = problems.lp_random_sparse(m=120, n=400, per_col=5, seed=13)
    A = sp.csc_matrix(A); A.sort_indices()
    kw = dict(eps=1e-5, restart_thresh=40, restart_fre=25)
    o = oracle_built.solve("oracle", A, b, c, linsys=linsys, **kw)
    plain = oracle_built.solve("oracle", A, b, c, linsys=linsys, eps=1e-5)
    assert o.info["admm_iter"] > 100 and o.info["admm_iter"] != plain.info["admm_iter"]
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, **kw) as S:
        info = S.solve()
        assert info["status_val"] == o.info["status_val"] and info["ipm_iter"] == o.info["ipm_iter"]
        assert info["admm_iter"] == o.info["admm_iter"], (linsys, info["admm_iter"], o.info["admm_iter"])
        for k in "xys":
            assert rel(getattr(S, k), getattr(o, k)) < 1e-6, (linsys, k)


def test_dense_tail_failure_falls_back_to_the_level_scheduled_factor(gpu, monkeypatch):
    """If the dense tail cannot be set up (no room for the two T x T triangles, a pivot the dense LDL' cannot take) abip_init
    re-factors without it instead of failing."""
    monkeypatch.delenv("ABIP_HIP_TAIL", raising=False)
    run_with_hooks("""
        z, A, b, c = load("lp_staircase")
        rng = np.random.default_rng(2)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=50) as S:
            assert S.scalar("tail") == 0 and S.scalar("levels_fwd") > 100
            Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
            rhs = rng.standard_normal(S.m + S.n)
            sol, its = S.kkt_solve(rhs, None, -1)
            assert rel(kkt_matrix(Asc, 1e-3) @ sol, rhs) < 1e-11
        """, {"ABIP_HIP_TAIL_FAIL": "1"})


def test_tail_residual_guard_falls_back(gpu, monkeypatch):
    """Set-up guard: one known right-hand side is solved with the factor at abip_init and ||K z - rhs|| checked on the
    host; a dense tail whose explicit inverse lost accuracy (test hook: the check reports failure) is replaced by the level-scheduled factor."""
    monkeypatch.delenv("ABIP_HIP_TAIL", raising=False)
    z, A, b, c = load("lp_staircase")
    with gpu.Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=50) as S:
        assert S.scalar("tail") > 0 and 0 <= S.scalar("factor_resid") < 1e-10
    monkeypatch.setenv("ABIP_HIP_TAIL_RESID_FAIL", "1")     # the shipped library ignores the hook ...
    with gpu.Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=50) as S:
        assert S.scalar("tail") > 0
    run_with_hooks("""
        z, A, b, c = load("lp_staircase")
        rng = np.random.default_rng(2)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=50) as S:
            assert S.scalar("tail") == 0 and S.scalar("levels_fwd") > 100 and 0 <= S.scalar("factor_resid") < 1e-10
            Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
            rhs = rng.standard_normal(S.m + S.n)
            sol, its = S.kkt_solve(rhs, None, -1)
            assert rel(kkt_matrix(Asc, 1e-3) @ sol, rhs) < 1e-11
        """, {"ABIP_HIP_TAIL_RESID_FAIL": "1"})                # ... the tests' variant takes it


def test_direct_solve_wide_head_with_tail(gpu, monkeypatch):
    """A factor whose head levels are wider than one workgroup (segmented path) and whose tail is several thousand pivots; the backward wide levels through the
    CSR-stream kernel (ABIP_HIP_TRI_LDS=0) and with the tail's solution resident in LDS (=1: dev_sptrsv.h k_tri_wide_lds, by itself only from 1e6 non-zeros in those
    levels on): the same solution to 1e-12."""
    from abip_amd import problems
    A, b, c = problems.lp_random_sparse(m=2000, n=10000, per_col=4)[:3]
    A = sp.csc_matrix(A); A.sort_indices()
    rhs = np.random.default_rng(8).standard_normal(A.shape[0] + A.shape[1])
    sols = {}
    for lds in ("0", "1"):
        monkeypatch.setenv("ABIP_HIP_TRI_LDS", lds)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0) as S:
            assert S.scalar("tail") >= 1024 and S.scalar("small_solve") == 0
            Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
            K = kkt_matrix(Asc, 1e-3)
            sol, its = S.kkt_solve(rhs, None, -1)
            assert rel(K @ sol, rhs) < 1e-10, lds
            sols[lds] = sol
    assert rel(sols["1"], sols["0"]) < 1e-12


def test_spmv_long_rows_and_ragged_blocks(gpu):
    """Rows longer than one LDS chunk (1024 non-zeros), empty rows of A' (empty columns are rejected upstream only with a
    warning) and 1-entry rows exercise every branch of the CSR-stream kernel."""
    rng = np.random.default_rng(3)
    m, n = 40, 6000
    dense_rows = sp.random(3, n, density=0.6, random_state=rng, format="csr")          # ~3600 nnz per row
    rest = sp.random(m - 3, n, density=0.002, random_state=rng, format="csr")
    A = sp.vstack([dense_rows, rest]).tocsc()
    A = sp.hstack([A, sp.identity(m, format="csc")], format="csc")
    b = A @ rng.random(A.shape[1]); c = rng.random(A.shape[1]) + 0.1
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0) as S:
        Asc = sp.csc_matrix((S.vector("Ax"), sp.csc_matrix(A).indices, sp.csc_matrix(A).indptr), shape=A.shape)
        x, y = rng.standard_normal(S.n), rng.standard_normal(S.m)
        assert rel(S.accum_by_A(x, np.zeros(S.m)), Asc @ x) < 1e-13
        assert rel(S.accum_by_Atrans(y, np.zeros(S.n)), Asc.T @ y) < 1e-13


def test_sliced_ell_layout(gpu, oracle_built, monkeypatch):
    """The SELL-64 image of A' (dev_common.h spmv_sell; built where natural-order slices pad little, e.g. the CSC columns of the C4 generator):
    products to 1e-14, ragged last slice and one-entry rows included, and the PCG trajectory still follows the oracle."""
    po = oracle_built
    monkeypatch.setenv("ABIP_HIP_SELL", "2")          # build it for small matrices too
    z, A, b, c = load("lp_random_sparse_small")
    rng = np.random.default_rng(4)
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-9) as S:
        assert S.scalar("sell_At") == (S.n + 63) // 64
        Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
        x, y = rng.standard_normal(S.n), rng.standard_normal(S.m)
        assert rel(S.accum_by_Atrans(y, x), x + Asc.T @ y) < 1e-14
        assert rel(S.accum_by_A(x, y), y + Asc @ x) < 1e-14
        T = 25
        o = po.solve("oracle", A, b, c, linsys="indirect", eps=1e-9, trace=T, max_admm_iters=100000)
        S.begin()
        for t in range(T):
            S.step(1)
            for col, nm in enumerate(("u", "v", "u_t")):
                assert rel(S.vector(nm), o.trace[t, col]) < 1e-9, (t + 1, nm)
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-6) as S:
        info = S.solve()
        _check_run_against_golden((info, S.x, S.y, S.s), z, "indirect_1e-06", 1e-6)
    monkeypatch.setenv("ABIP_HIP_SELL", "0")
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0) as S:
        assert S.scalar("sell_At") == 0


@pytest.mark.parametrize("name,eps", [("lp_random_sparse_small", 1e-6), ("lp_multicommodity_small", 1e-4)])
def test_streamed_launch_path_is_the_stepwise_one(gpu, name, eps, monkeypatch):
    """The launch path of the PCG back-end (the C4 path) enqueues iteration j + 1 before it has read the verdict of iteration j, keeps A'u_y across an iteration,
    streams the Barzilai-Borwein search with its decisions on the device and hands a look-ahead's second step to the next one when the penalty did not change
    (solver.hip: admm_stream_pcg, adaptive_search_stream).  None of that may change a bit of the ITERATE (with ABIP_HIP_ATY the dual residual sums of the stopping
    test are dealt to the workgroups differently -- equal to rounding, dev_kernels.h k_q_A_aty -- which could only show as a count that differs at an exact tie:
    the counts are asserted equal here): every combination of the switches -- and a blind PCG count forced so
    small that every iteration and every look-ahead stalls and is resumed (ABIP_HIP_STREAM_BLIND=2) -- ends on the iterate, the counts and the PCG total of
    round 4's form (one control read per iteration, every product where the reference forms it), and on the reference's fixture."""
    monkeypatch.setenv("ABIP_HIP_XCD", "0")
    z, A, b, c = load(name)
    runs = {}
    variants = {"round4": dict(ABIP_HIP_ATY="0", ABIP_HIP_STREAM="0", ABIP_HIP_STREAM_BB="0"), "aty": dict(ABIP_HIP_ATY="1", ABIP_HIP_STREAM="0", ABIP_HIP_STREAM_BB="0"),
                "stream": dict(ABIP_HIP_ATY="0", ABIP_HIP_STREAM="1", ABIP_HIP_STREAM_BB="0"), "search": dict(ABIP_HIP_ATY="1", ABIP_HIP_STREAM="1", ABIP_HIP_STREAM_BB="1", ABIP_HIP_BB_REUSE="0"),
                "default": dict(ABIP_HIP_ATY="1", ABIP_HIP_STREAM="1", ABIP_HIP_STREAM_BB="1", ABIP_HIP_BB_REUSE="1"),
                "stalls": dict(ABIP_HIP_ATY="1", ABIP_HIP_STREAM="1", ABIP_HIP_STREAM_BB="1", ABIP_HIP_BB_REUSE="1", ABIP_HIP_STREAM_BLIND="2")}
    for vn, env in variants.items():
        for k in ("ABIP_HIP_ATY", "ABIP_HIP_STREAM", "ABIP_HIP_STREAM_BB", "ABIP_HIP_BB_REUSE", "ABIP_HIP_STREAM_BLIND"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=eps) as S:
            assert S.scalar("xcd") == 0.0
            info = S.solve()
            runs[vn] = (info["ipm_iter"], info["admm_iter"], S.scalar("tot_cg_its"), info["pobj"], S.x.copy(), S.y.copy(), S.s.copy(), S.scalar("stream_stalls"), S.scalar("stream_iters"))
            if vn == "round4":
                _check_against_golden(S, info, z, f"indirect_{eps:g}", eps, name=name)
    ref = runs["round4"]
    assert ref[7] == 0 and ref[8] == 0 and runs["default"][8] > 0 and runs["stalls"][7] > runs["stalls"][8] > 0      # (round 4's form streams nothing; the forced run stalls more than once per iteration)
    for vn, r in runs.items():
        assert r[:4] == ref[:4], (vn, r[:4], ref[:4])
        for a2, b2 in zip(r[4:7], ref[4:7]):
            assert np.array_equal(a2, b2), vn


# ---------------------------------------------------------------------------------------------- trajectories
@pytest.mark.parametrize("name", ["lp_afiro_like", "lp_random_sparse_small", "lp_multicommodity_small"])
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_iterates_follow_the_oracle(gpu, oracle_built, name, linsys):
    po = oracle_built
    z, A, b, c = load(name)
    T = 25
    o = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-9, trace=T, max_admm_iters=100000)
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-9) as S:
        S.begin()
        assert rel(S.vector("g"), o.work["g"]) < 1e-10 and abs(S.scalar("g_th") - o.work["g_th"]) < 1e-10 * abs(o.work["g_th"])
        for t in range(min(T, len(o.trace))):
            fin, done = S.step(1)
            assert done == 1 and not fin
            for col, nm in enumerate(("u", "v", "u_t")):
                assert rel(S.vector(nm), o.trace[t, col]) < 1e-9, (name, linsys, t + 1, nm)


# ---------------------------------------------------------------------------------------------- full solves
# The one fixture on which a device path does NOT reproduce the reference's iteration counts (profiles/r04_parity_counts.txt lists every fixture x eps x
# back-end x path; scripts/parity_counts.py).  lp_tiny_scale5 (60 x 150, scale = 5) is degenerate in x: the reference's own eps = 1e-4 stopping point is 9.8e-2
# away (relative) from the x it converges to at eps = 1e-8, while its objective is already right to 1e-5 -- and its trajectory is not determined beyond the first
# few outer iterations: profiles/r05c_knife_edge_trace_*.txt follows oracle and device side by side; a difference of 3e-16 in the iterate after the first
# Barzilai-Borwein search is 1.4e-11 in beta, 3e-8 five searches later, and by the 13th search beta is 2.94 against 3.00 (the search divides inner products of
# differences of nearly equal vectors, adaptive.c:154-229).  On it the persistent launch -- which regroups sums the launch path does not (the dense inverse of
# rho I + A A' where the reference solves with LDL'; u_t'h formed as y'(h_y + A h_x) - rhs_x'h_x so that it rides on the back-substitution's exchange) -- lands on
# the other side of a Barzilai-Borwein decision (direct: 15 or 16 outer iterations, 264 / 287 / 293 inner ones; PCG: 16 / 275 against 282).  There the bar is what
# the problem determines: status, objective, the reference's convergence criteria -- and (x, y, s) against the reference's own eps = 1e-8 solution
# (test_knife_edge_fixture_at_tight_eps).  The exemption is the persistent launch's alone: the launch path takes the reference's counts here too and is held to them.
KNIFE_EDGE = {("lp_tiny_scale5", "direct_0.0001"), ("lp_tiny_scale5", "indirect_0.0001")}


def _check_against_golden(S, info, z, tag, eps, name=None):
    """A run stopped at tolerance eps is only defined up to O(eps): the PCG stopping test (indirect.c:375) and the inner
    stopping test (abip.c:2173) are discontinuous in sums the device forms in a different order.  Measured (profiles/r04_parity_counts.txt):
    identical outer AND inner iteration counts on every fixture but the one named in KNIFE_EDGE -- so the counts are asserted equal, and
    (x, y, s) and the objectives within 10*eps of the reference's, with the reference's own convergence criteria satisfied.  Agreement to
    1e-6 is checked where it is meaningful: at tight eps (test_tight_tolerance_agreement)."""
    g = info_of(z, tag)
    assert info["status_val"] == g["status_val"]
    tol = 10 * eps
    if (name, tag) in KNIFE_EDGE and S.scalar("xcd") == 1.0:
        assert abs(info["ipm_iter"] - g["ipm_iter"]) <= 1 and abs(info["admm_iter"] - g["admm_iter"]) <= 0.12 * g["admm_iter"]
    else:
        assert info["ipm_iter"] == g["ipm_iter"] and info["admm_iter"] == g["admm_iter"], (tag, info["ipm_iter"], info["admm_iter"], g["ipm_iter"], g["admm_iter"])
        for k in "xys":
            assert rel(getattr(S, k), z[f"{tag}_{k}"]) < tol, (tag, k, info["admm_iter"], g["admm_iter"])
    assert abs(info["pobj"] - g["pobj"]) <= tol * (1 + abs(g["pobj"]))
    assert abs(info["dobj"] - g["dobj"]) <= tol * (1 + abs(g["dobj"]))
    for k in ("res_pri", "res_dual", "rel_gap"):
        assert info[k] < eps


def _check_run_against_golden(run, z, tag, eps):
    """The same bar for a run handed over as (info, x, y, s): the reference's counts, (x, y, s) and objectives within 10 eps."""
    info, x, y, s_ = run
    g = info_of(z, tag)
    assert info["status_val"] == g["status_val"]
    assert (info["ipm_iter"], info["admm_iter"]) == (g["ipm_iter"], g["admm_iter"]), (tag, info["ipm_iter"], info["admm_iter"], g["ipm_iter"], g["admm_iter"])
    assert abs(info["pobj"] - g["pobj"]) <= 10 * eps * (1 + abs(g["pobj"]))
    for got, k in ((x, "x"), (y, "y"), (s_, "s")):
        assert rel(got, z[f"{tag}_{k}"]) < 10 * eps, k


@pytest.mark.parametrize("name", ["lp_afiro_like", "lp_random_sparse_small", "lp_multicommodity_small"])
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_tight_tolerance_agreement(gpu, oracle_built, name, linsys):
    """north_star: converge to the reference's (x, y, s) within 1e-6 relative.  Both sides run to eps = 1e-8 (the PCG
    back-end cannot go lower: its tolerance is floored at 1e-7 absolute, indirect.c:409 -- reference and device alike
    stall at eps = 1e-9)."""
    po = oracle_built
    z, A, b, c = load(name)
    o = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-8, max_admm_iters=300000)
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-8, max_admm_iters=300000) as S:
        info = S.solve()
        assert info["status_val"] == o.info["status_val"] == 1
        for k in "xys":
            assert rel(getattr(S, k), getattr(o, k)) < 1e-6, (name, linsys, k)
        assert abs(info["pobj"] - o.info["pobj"]) <= 1e-6 * (1 + abs(o.info["pobj"]))
        assert abs(info["pobj"] - info["dobj"]) <= 1e-6 * (1 + abs(info["pobj"]))


@pytest.mark.parametrize("name", ["lp_afiro_like", "lp_random_sparse_small", "lp_multicommodity_small", "lp_staircase"])
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_north_star_agreement_with_the_reference_itself(gpu, name, linsys):
    """north_star bar, against the REAL reference: both run to eps = 1e-8 (fixtures <linsys>_1e-08_* written by make_golden.py from
    oracle/_ref); the device's (x, y, s) and objectives must agree with the reference's to 1e-6 relative."""
    z, A, b, c = load(name)
    tag = f"{linsys}_1e-08"
    g = info_of(z, tag)
    assert g["status_val"] == 1
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-8, max_admm_iters=400000) as S:
        info = S.solve()
        assert info["status_val"] == 1
        assert (info["ipm_iter"], info["admm_iter"]) == (g["ipm_iter"], g["admm_iter"]), (name, linsys, info["ipm_iter"], info["admm_iter"], g["ipm_iter"], g["admm_iter"])
        for k in "xys":   # (lp_staircase -- the C2 workload, a degenerate LP -- included: profiles/r04w_parity_counts.txt has it at 3e-9)
            assert rel(getattr(S, k), z[f"{tag}_{k}"]) < 1e-6, (name, linsys, k)
        assert abs(info["pobj"] - g["pobj"]) <= 1e-6 * (1 + abs(g["pobj"]))
        assert abs(info["dobj"] - g["dobj"]) <= 1e-6 * (1 + abs(g["dobj"]))
        for k in ("res_pri", "res_dual", "rel_gap"):
            assert info[k] < 1e-8


@pytest.mark.parametrize("xcd", ["1", "0"])
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
@pytest.mark.parametrize("name,eps_list", [("lp_afiro_like", (1e-3, 1e-6, 1e-8)), ("lp_random_sparse_small", (1e-3, 1e-6, 1e-8)),
                                           ("lp_multicommodity_small", (1e-4, 1e-8)), ("lp_staircase", (1e-3, 1e-6))])
def test_exact_iteration_counts_on_both_paths(gpu, monkeypatch, name, eps_list, linsys, xcd):
    """Every main fixture x eps x back-end, through the persistent launch (PCG inside the kernel; the direct variant's dense inverse of rho I + A A') and through
    the launch-per-operation path (the C4 path: PCG kernel by kernel; the sparse LDL' solve): the reference's EXACT inner and outer iteration counts, (x, y, s) and
    the objectives within 10 eps, its convergence criteria (profiles/r04w_parity_counts.txt is this table).  lp_staircase is the C2 workload.
    (The launch path skips the two longest PCG runs -- 40 000 iterations of lp_multicommodity_small / lp_staircase at 1e-8, ~18 s each; the persistent launch and
    test_north_star_agreement_with_the_reference_itself cover those.)"""
    monkeypatch.setenv("ABIP_HIP_XCD", xcd)
    z, A, b, c = load(name)
    for eps in eps_list:
        if xcd == "0" and linsys == "indirect" and eps == 1e-8 and name == "lp_multicommodity_small":
            continue
        with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=eps, max_admm_iters=400000) as S:
            assert S.scalar("xcd") == float(xcd)
            info = S.solve()
            _check_against_golden(S, info, z, f"{linsys}_{eps:g}", eps, name=name)


@pytest.mark.parametrize("xcd", ["1", "0"])
@pytest.mark.parametrize("variant", sorted(TINY_VARIANTS))
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_non_default_switches_match_reference_fixture(gpu, variant, linsys, xcd, monkeypatch):
    """half_update, origin / qp scaling, no normalisation, no adaptive search, scale = 5, the "tedious" mu table: both device paths at the reference's counts
    (the persistent launch on the KNIFE_EDGE fixture excepted; the launch path is held to it there too)."""
    monkeypatch.setenv("ABIP_HIP_XCD", xcd)
    z, A, b, c = load("lp_tiny_" + variant)
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-4, **TINY_VARIANTS[variant]) as S:
        assert S.scalar("xcd") == float(xcd)
        info = S.solve()
        _check_against_golden(S, info, z, f"{linsys}_0.0001", 1e-4, name="lp_tiny_" + variant)


def test_knife_edge_fixture_at_tight_eps(gpu, monkeypatch):
    """lp_tiny_scale5 (KNIFE_EDGE above) against the REFERENCE's eps = 1e-8 solution (fixture tags <linsys>_1e-08_*, written by make_golden.py from oracle/_ref),
    both back-ends, the three device paths -- launch path, persistent launch in batches and spanning outer iterations.  What this LP determines is held to the
    north-star bar: status, (x, y, s) to 1e-6, the objectives to 1e-8, the reference's convergence criteria.  What it does not determine is its iteration count:
    profiles/r05c_knife_edge_trace_*.txt follows oracle and device outer iteration by outer iteration -- after the FIRST Barzilai-Borwein search the iterates agree
    to 3e-16 and the penalties beta to 1.4e-11 (the search divides inner products of differences of nearly equal vectors, adaptive.c:154-229), five searches later to
    3e-8, by the 13th beta is 2.94 against 3.00 and the 14th 2.02 against 3.87: any other summation order than the reference's own ends elsewhere (15 or 16 outer
    iterations, 1832 / 1833-1842 inner ones with LDL'; 1964 / 1954 with PCG on the launch path, 2004 in the persistent launch's batch form).  Hence +-1 outer and 12 % inner
    iterations here -- the KNIFE_EDGE bar of eps 1e-4 -- and only here."""
    z, A, b, c = load("lp_tiny_scale5")
    for linsys in ("direct", "indirect"):
        tag = f"{linsys}_1e-08"
        g = info_of(z, tag)
        assert g["status_val"] == 1
        for mode, env in (("path", {"ABIP_HIP_XCD": "0"}), ("batch", {"ABIP_HIP_XCD": "1", "ABIP_HIP_XCD_OUTER": "0"}), ("whole", {"ABIP_HIP_XCD": "1", "ABIP_HIP_XCD_OUTER": "1"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-8, **TINY_VARIANTS["scale5"]) as S:
                info = S.solve()
                assert info["status_val"] == 1, (linsys, mode)
                assert abs(info["ipm_iter"] - g["ipm_iter"]) <= 1 and abs(info["admm_iter"] - g["admm_iter"]) <= 0.12 * g["admm_iter"], (linsys, mode, info["ipm_iter"], info["admm_iter"])
                for k in "xys":
                    assert rel(getattr(S, k), z[f"{tag}_{k}"]) < 1e-6, (linsys, mode, k)
                assert abs(info["pobj"] - g["pobj"]) <= 1e-8 * (1 + abs(g["pobj"])) and abs(info["dobj"] - g["dobj"]) <= 1e-8 * (1 + abs(g["dobj"]))
                for k in ("res_pri", "res_dual", "rel_gap"):
                    assert info[k] < 1e-8


def test_matlab_surface_end_to_end(gpu):
    z, A, b, c = load("lp_afiro_like")
    for pcg, tag in ((0, "direct_1e-06"), (1, "indirect_1e-06")):
        p = gpu.abip_get_params()
        p.update(verbose=0, pcg=pcg, tol=1e-6)
        x, y, s, info = gpu.abip(dict(A=A, b=b, c=c), {"l": A.shape[1]}, p)
        assert info["status"] == "Solved" and info["solver"] == "abip-lp"
        assert rel(x, z[tag + "_x"]) < 1e-6 and rel(y, z[tag + "_y"]) < 1e-6 and rel(s, z[tag + "_s"]) < 1e-6
        assert abs(info["pobj"] - float(c @ x)) < 1e-12


# ---------------------------------------------------------------------------------------------- edge cases
def test_status_codes_on_infeasible_and_unbounded(gpu, oracle_built):
    po = oracle_built
    cases = {
        "infeasible": (sp.csc_matrix(np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 1.0]])), np.array([-1.0, 2.0]), np.array([1.0, 1.0, 1.0])),
        "unbounded": (sp.csc_matrix(np.array([[1.0, -1.0, 0.0], [0.0, 1.0, -1.0]])), np.array([0.0, 0.0]), np.array([-1.0, 0.0, 0.0])),
    }
    for nm, (A, b, c) in cases.items():
        for linsys in ("direct", "indirect"):
            o = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-5, max_admm_iters=20000)
            with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-5, max_admm_iters=20000) as S:
                info = S.solve()
            assert info["status_val"] == o.info["status_val"], (nm, linsys, info["status"], o.info["status"])
            assert info["status"] == o.info["status"]


def test_iteration_limits_and_inaccurate_status(gpu, oracle_built):
    po = oracle_built
    z, A, b, c = load("lp_random_sparse_small")
    for linsys in ("indirect", "direct"):
        o = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-9, max_admm_iters=60)
        with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-9, max_admm_iters=60) as S:
            info = S.solve()
            assert info["status"] == o.info["status"] == "Solved/Inaccurate"
            assert info["admm_iter"] == o.info["admm_iter"] and info["ipm_iter"] == o.info["ipm_iter"]
            assert rel(S.x, o.x) < 1e-8 and rel(S.y, o.y) < 1e-8 and rel(S.s, o.s) < 1e-8


def test_invalid_input_is_rejected_like_the_reference(gpu):
    A = sp.csc_matrix(np.ones((3, 2)))                       # m > n  (abip.c:1661-1665)
    with pytest.raises(RuntimeError):
        gpu.Solver(A, np.ones(3), np.ones(2), verbose=0)
    A = sp.identity(3, format="csc")
    with pytest.raises(RuntimeError):
        gpu.Solver(A, np.ones(3), np.ones(3), verbose=0, alpha=2.5)   # alpha must be in (0,2)


def test_warm_start_quirk_matches_oracle(gpu, oracle_built):
    """warm_start_vars overwrites the guess with sqrt(mu/beta) (abip.c:328-347): same iterates as the reference anyway."""
    po = oracle_built
    z, A, b, c = load("lp_afiro_like")
    warm = (z["indirect_1e-06_x"], z["indirect_1e-06_y"], z["indirect_1e-06_s"])
    o = po.solve("oracle", A, b, c, linsys="indirect", eps=1e-5, warm=warm, warm_start=1, max_admm_iters=400)
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-5, warm_start=1, max_admm_iters=400) as S:
        S.x[:], S.y[:], S.s[:] = warm
        info = S.solve()
        assert info["admm_iter"] == o.info["admm_iter"] and info["status"] == o.info["status"] and rel(S.x, o.x) < 1e-7


# ---------------------------------------------------------------------------------------------- full size (C4)
def test_full_size_properties(gpu):
    """BASELINE configs[3] shape (m=200k, n=500k, nnz~5M): adjoint identity <Ax, y> = <x, A'y>, linearity, a KKT solve
    whose residual is checked with the device SpMVs themselves, and monotone decrease of the barrier parameter."""
    from abip_amd import problems
    A, b, c = problems.lp_random_sparse()
    rng = np.random.default_rng(7)
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-6) as S:
        m, n = S.m, S.n
        x1, x2, y = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(m)
        Ax1 = S.accum_by_A(x1, np.zeros(m)); Ax2 = S.accum_by_A(x2, np.zeros(m)); Aty = S.accum_by_Atrans(y, np.zeros(n))
        assert abs(Ax1 @ y - x1 @ Aty) <= 1e-10 * (np.linalg.norm(Ax1) * np.linalg.norm(y))
        assert rel(S.accum_by_A(2.0 * x1 - 3.0 * x2, np.zeros(m)), 2.0 * Ax1 - 3.0 * Ax2) < 1e-13
        rhs = rng.standard_normal(m + n)
        sol, its = S.kkt_solve(rhs, None, -1)
        ry = 1e-3 * sol[:m] + S.accum_by_A(sol[m:], np.zeros(m)) - rhs[:m]
        rx = S.accum_by_Atrans(sol[:m], np.zeros(n)) - sol[m:] - rhs[m:]
        assert its > 0 and np.sqrt(ry @ ry + rx @ rx) / np.linalg.norm(rhs) < 1e-6
        S.begin()
        mus = []
        for _ in range(6):
            S.step(5)
            mus.append(S.scalar("mu"))
        assert all(b2 <= a2 for a2, b2 in zip(mus, mus[1:])) and mus[-1] < 1.0
        info = S.end()
        assert np.isfinite(info["res_pri"]) and np.isfinite(info["res_dual"])


# ---------------------------------------------------------------------------------------------- full size (C3)
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_full_size_c3_properties(gpu, linsys):
    """BASELINE configs[2] surrogate at full size (pds-class multi-commodity LP, 16 390 x 48 400), both back-ends: adjoint identity and
    linearity of the device products, a KKT solve checked with the device SpMVs, and a solve to eps 1e-4 whose answer satisfies the
    reference's own optimality criteria (primal / dual residual and gap recomputed on the host from the returned (x, y, s))."""
    from abip_amd import problems
    A, b, c = problems.lp_multicommodity(nodes=1200, arcs=4400, commodities=10)[:3]
    A = sp.csc_matrix(A)
    rng = np.random.default_rng(11)
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-4) as S:
        m, n = S.m, S.n
        assert (m, n) == A.shape and m > 16000 and n > 48000
        x1, x2, y = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(m)
        Ax1 = S.accum_by_A(x1, np.zeros(m)); Ax2 = S.accum_by_A(x2, np.zeros(m)); Aty = S.accum_by_Atrans(y, np.zeros(n))
        assert abs(Ax1 @ y - x1 @ Aty) <= 1e-10 * (np.linalg.norm(Ax1) * np.linalg.norm(y))
        assert rel(S.accum_by_A(2.0 * x1 - 3.0 * x2, np.zeros(m)), 2.0 * Ax1 - 3.0 * Ax2) < 1e-13
        rhs = rng.standard_normal(m + n)
        sol, its = S.kkt_solve(rhs, None, -1)
        ry = 1e-3 * sol[:m] + S.accum_by_A(sol[m:], np.zeros(m)) - rhs[:m]
        rx = S.accum_by_Atrans(sol[:m], np.zeros(n)) - sol[m:] - rhs[m:]
        assert np.sqrt(ry @ ry + rx @ rx) / np.linalg.norm(rhs) < (1e-6 if linsys == "indirect" else 1e-9)
        info = S.solve()
        assert info["status_val"] == 1
        x, yv, s_ = S.x, S.y, S.s
        assert np.linalg.norm(A @ x - b) / (1 + np.linalg.norm(b)) < 1.5e-4
        assert np.linalg.norm(A.T @ yv + s_ - c) / (1 + np.linalg.norm(c)) < 1.5e-4
        assert abs(c @ x - b @ yv) / (1 + abs(c @ x) + abs(b @ yv)) < 1.5e-4
        assert x.min() > -1e-6 and s_.min() > -1e-6
