"""CPU, dev container only: the oracle against the live reference build (oracle/_ref) on fresh seeds.
Skipped where oracle/_ref is absent."""
import numpy as np
import pytest

from _golden import rel
from abip_amd import problems


@pytest.mark.parametrize("seed", [11, 12])
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_random_lp_against_live_reference(oracle_built, seed, linsys):
    po = oracle_built
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    A, b, c = problems.lp_random_sparse(m=120, n=400, per_col=5, seed=seed)
    r = po.solve("ref", A, b, c, linsys=linsys, eps=1e-5)
    o = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-5)
    assert r.info["status_val"] == o.info["status_val"] == 1
    assert r.info["admm_iter"] == o.info["admm_iter"] and r.info["ipm_iter"] == o.info["ipm_iter"]
    tol = 1e-12 if linsys == "indirect" else 1e-7
    for k in "xys":
        assert rel(getattr(o, k), getattr(r, k)) < tol
    assert r.settings_after == o.settings_after   # the solver mutates caller-owned settings identically


@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_restart_path_against_live_reference(oracle_built, linsys):
    """restart_vars (abip.c:587-630): with the threshold lowered from 1e5 to 40 iterations and a period of 25 the periodic
    restart from the running mean fires many times inside one solve."""
    po = oracle_built
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    A, b, c = problems.lp_random_sparse(m=120, n=400, per_col=5, seed=13)
    kw = dict(linsys=linsys, eps=1e-5, restart_thresh=40, restart_fre=25)
    r = po.solve("ref", A, b, c, **kw)
    o = po.solve("oracle", A, b, c, **kw)
    plain = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-5)
    assert r.info["admm_iter"] > 100 and r.info["admm_iter"] != plain.info["admm_iter"]      # the restarts did change the run
    assert r.info["status_val"] == o.info["status_val"]
    assert r.info["admm_iter"] == o.info["admm_iter"] and r.info["ipm_iter"] == o.info["ipm_iter"]
    tol = 1e-12 if linsys == "indirect" else 1e-7
    for k in "xys":
        assert rel(getattr(o, k), getattr(r, k)) < tol


def _dense_lp(m, n, density, seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    A = sp.csc_matrix(A + sp.hstack([sp.identity(m), sp.csc_matrix((m, n - m))]))
    return A, A @ (rng.random(n) + 0.1), rng.random(n) + 0.1


@pytest.mark.parametrize("linsys", ["indirect", "direct"])
@pytest.mark.parametrize("shape", [(12, 40, 0.3, 1), (10, 30, 0.9, 2), (30, 34, 0.2, 3)])
def test_sparsity_dependent_inner_caps_against_live_reference(oracle_built, shape, linsys):
    """abip.c:2104-2115: the inner-iteration cap depends on the sparsity of A (sp > 0.5, 0.2 < sp <= 0.5); dense little LPs reach both."""
    po = oracle_built
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    A, b, c = _dense_lp(*shape)
    r = po.solve("ref", A, b, c, linsys=linsys, eps=1e-6, max_admm_iters=100000)
    o = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-6, max_admm_iters=100000)
    assert r.info["status_val"] == o.info["status_val"]
    assert r.info["admm_iter"] == o.info["admm_iter"] and r.info["ipm_iter"] == o.info["ipm_iter"]
    for k in "xys":
        assert rel(getattr(o, k), getattr(r, k)) < (1e-12 if linsys == "indirect" else 1e-7)


@pytest.mark.parametrize("linsys", ["indirect", "direct"])
@pytest.mark.parametrize("variant", ["half", "origin", "qp", "nonorm", "noadapt", "scale5", "tedious"])
def test_non_default_switches_against_live_reference(oracle_built, variant, linsys):
    """Every non-default algorithm switch the mex gateway exposes (abip_mex.c:183-341), on a fresh seed: the oracle must walk the reference's path
    iteration for iteration (the committed fixtures pin the same switches on one tiny LP only)."""
    from _golden import TINY_VARIANTS
    po = oracle_built
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    A, b, c = problems.lp_random_sparse(m=60, n=200, per_col=4, seed=21)
    kw = dict(linsys=linsys, eps=1e-4, **TINY_VARIANTS[variant])
    r = po.solve("ref", A, b, c, **kw)
    o = po.solve("oracle", A, b, c, **kw)
    assert r.info["status_val"] == o.info["status_val"]
    assert r.info["admm_iter"] == o.info["admm_iter"] and r.info["ipm_iter"] == o.info["ipm_iter"]
    for k in "xys":
        # direct: the oracle orders the KKT matrix differently from SuiteSparse AMD, so the solves round differently (1e-16) and the ADMM map carries that forward
        assert rel(getattr(o, k), getattr(r, k)) < (1e-11 if linsys == "indirect" else 5e-6)


@pytest.mark.parametrize("linsys", ["indirect", "direct"])
@pytest.mark.parametrize("kind", ["infeasible", "unbounded"])
def test_certificate_statuses_against_live_reference(oracle_built, kind, linsys):
    """Status logic of get_solution (abip.c:1374-1404) on an infeasible and an unbounded LP: same status code, same iteration counts as the reference."""
    import scipy.sparse as sp
    po = oracle_built
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    if kind == "infeasible":
        A, b, c = sp.csc_matrix(np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 1.0]])), np.array([-1.0, 2.0]), np.array([1.0, 1.0, 1.0])
    else:
        A, b, c = sp.csc_matrix(np.array([[1.0, -1.0, 0.0], [0.0, 1.0, -1.0]])), np.array([0.0, 0.0]), np.array([-1.0, 0.0, 0.0])
    kw = dict(linsys=linsys, eps=1e-5, max_admm_iters=20000)
    r = po.solve("ref", A, b, c, **kw)
    o = po.solve("oracle", A, b, c, **kw)
    assert r.info["status_val"] == o.info["status_val"] and r.info["status_val"] != 1
    assert r.info["admm_iter"] == o.info["admm_iter"] and r.info["ipm_iter"] == o.info["ipm_iter"]


@pytest.mark.parametrize("linsys", ["indirect", "direct"])
@pytest.mark.parametrize("gen", ["staircase", "multicommodity"])
def test_structured_lps_against_live_reference(oracle_built, gen, linsys):
    """The structured generators behind the C2 / C3 surrogates at sizes the reference finishes in seconds, other seeds than the committed fixtures."""
    po = oracle_built
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    A, b, c = problems.lp_staircase(seed=5, stages=5, rows_per=20, cols_per=44) if gen == "staircase" else problems.lp_multicommodity(seed=3, nodes=24, arcs=80, commodities=3)
    r = po.solve("ref", A, b, c, linsys=linsys, eps=1e-4)
    o = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-4)
    assert r.info["status_val"] == o.info["status_val"]
    assert r.info["admm_iter"] == o.info["admm_iter"] and r.info["ipm_iter"] == o.info["ipm_iter"]
    for k in "xys":
        assert rel(getattr(o, k), getattr(r, k)) < (1e-10 if linsys == "indirect" else 1e-6)
