"""GPU (-m gpu): the LASSO reformulation (settings.prob_type 0, abip_ml surface) on the device against the CPU oracle's
restatement of lasso_config.c on the same inputs, and against the LASSO minimiser itself (scikit-learn)."""
import numpy as np
import pytest

from _lasso_cases import CASES, gen, objective

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


@pytest.mark.parametrize("eps", [1e-3, 1e-6])
@pytest.mark.parametrize("name", list(CASES))
def test_lasso_matches_the_oracle(gpu, pq, name, eps):
    """Same trajectory: the device runs the materialised operator through the KKT LDL', the oracle the matrix-free operator through
    the reduced Cholesky system (lasso_config.c:652-708): equal outer/inner iteration counts (+-1 inner iteration per outer one where a
    stopping metric sits on its threshold) and beta to 1e-6 relative."""
    X, y, lam = gen(name)
    want, wi = pq.solve_lasso(X, y, lam, eps=eps, eps_p=eps, eps_d=eps, eps_g=eps, eps_inf=eps, eps_unb=eps)
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, eps=eps, linsys_solver=1, verbose=0))
    assert info["status"] == wi["status"] == "Solved"
    assert info["ipm_iter"] == wi["ipm_iter"]
    assert abs(info["admm_iter"] - wi["admm_iter"]) <= max(2, wi["ipm_iter"])
    same = info["admm_iter"] == wi["admm_iter"]
    tol = 1e-7 if same else 10 * eps
    assert np.max(np.abs(sol["x"] - want)) <= tol * max(1.0, np.abs(want).max())
    assert abs(info["pobj"] - wi["pobj"]) <= tol * max(1.0, abs(wi["pobj"])) and abs(info["dobj"] - wi["dobj"]) <= tol * max(1.0, abs(wi["dobj"]))
    if same:
        for k in ("res_pri", "res_dual", "gap"):
            a, b = info[k], wi["rel_gap" if k == "gap" else k]
            assert abs(a - b) <= 1e-6 * max(abs(b), eps)


@pytest.mark.parametrize("linsys", [1, 3])
def test_lasso_reaches_the_minimiser(gpu, linsys):
    """Both KKT back-ends (LDL' and the y-space PCG of qcp_pcg.h) against scikit-learn's coordinate descent."""
    from sklearn.linear_model import Lasso
    X, y, lam = gen("wide_sparse_big")
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, eps=1e-5, linsys_solver=linsys, verbose=0))
    assert info["status"] == "Solved"
    sk = Lasso(alpha=lam / X.shape[0], fit_intercept=False, tol=1e-13, max_iter=200000).fit(X.toarray(), y).coef_
    f, fs = objective(X, y, lam, sol["x"]), objective(X, y, lam, sk)
    assert abs(f - fs) <= 5e-5 * max(1.0, abs(fs))
    assert np.max(np.abs(sol["x"] - sk)) <= 5e-3 * max(1.0, np.abs(sk).max())
    if linsys == 3:
        assert info["avg_cg_iters"] > 0


def test_lasso_surface_errors(gpu):
    X, y, lam = gen("wide_dense")
    with pytest.raises(ValueError):
        gpu.abip_ml(dict(X=X.toarray(), y=y, **{"lambda": lam}), dict(prob_type=0))
    with pytest.raises(ValueError):
        gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict())
    with pytest.raises(ValueError):
        gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=2))
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": 0.0}), dict(prob_type=0, verbose=0))
    assert info["status"] == "Failure" and info["status_val"] == -4
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, normalize=0, verbose=0))
    assert info["status"] == "Failure"


def test_lasso_at_the_reference_protocol_size(gpu, pq):
    """scripts/bench-qcp/test_lasso.m's smallest size (1000 x 5000, density 0.15, unit noise, eps 1e-3) through the front end on the device against the oracle's
    restatement: the same iteration counts (a few dozen inner iterations -- the formulation's scaling is tuned to this protocol) and beta."""
    from abip_amd import problems
    X, y, lam = problems.lasso_protocol_data(1000, 5000)
    want, wi = pq.solve_lasso(X, y, lam)
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, eps=1e-3, linsys_solver=1, verbose=0))
    assert info["status"] == wi["status"] == "Solved" and info["ipm_iter"] == wi["ipm_iter"] and info["admm_iter"] == wi["admm_iter"] and info["admm_iter"] < 100
    assert np.max(np.abs(sol["x"] - want)) <= 1e-8 * max(1.0, np.abs(want).max())
    solp, infop = gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, eps=1e-3, linsys_solver=3, verbose=0))
    assert infop["status"] == "Solved" and abs(infop["pobj"] - wi["pobj"]) <= 1e-3 * abs(wi["pobj"])
