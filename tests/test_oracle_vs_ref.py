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
