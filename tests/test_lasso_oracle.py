"""CPU: the LASSO reformulation of the oracle (oracle/abip_qcp_oracle.c, prob_type 0; restates lasso_config.c).
No reference build exists for src/abip-qcp (MKL headers, see the oracle's header) and the reference holds no expected
values for it, so what pins the restatement is the problem itself: the returned beta must be the LASSO minimiser, checked
against scikit-learn's coordinate descent (an independent solver), and the reported objectives / residuals must be the
ones the definitions of lasso_config.c:358-460 give when recomputed here from (X, y, lambda, beta)."""
import numpy as np
import pytest

from _lasso_cases import CASES, gen, objective


@pytest.fixture(scope="module")
def pq():
    from oracle import pyoracle_qcp
    pyoracle_qcp.lib()
    return pyoracle_qcp


@pytest.mark.parametrize("name", [k for k in CASES if k != "wide_sparse_big"])
def test_lasso_reformulation_reaches_the_lasso_minimiser(pq, name):
    from sklearn.linear_model import Lasso
    X, y, lam = gen(name)
    m = X.shape[0]
    beta, info = pq.solve_lasso(X, y, lam, eps=1e-5, eps_p=1e-5, eps_d=1e-5, eps_g=1e-5)
    assert info["status"] == "Solved"
    sk = Lasso(alpha=lam / m, fit_intercept=False, tol=1e-13, max_iter=200000).fit(X.toarray(), y).coef_
    f, fs = objective(X, y, lam, beta), objective(X, y, lam, sk)
    assert abs(f - fs) <= 2e-5 * max(1.0, abs(fs))
    assert np.max(np.abs(beta - sk)) <= 2e-3 * max(1.0, np.abs(sk).max())
    # pobj as defined at lasso_config.c:446-447 is the LASSO objective of the (slightly infeasible) conic iterate
    assert abs(info["pobj"] - fs) <= 1e-4 * max(1.0, abs(fs))
    assert info["res_pri"] < 1e-5 and info["res_dual"] < 1e-5 and info["rel_gap"] < 1e-5


def test_lasso_default_tolerance_and_iteration_counts_are_stable(pq):
    """A regression pin of the restatement itself (eps = 1e-3, the protocol of scripts/bench-qcp/test_lasso.m:11,84)."""
    got = {}
    for name in ("wide_dense", "tall_dense", "wide_sparse", "tall_sparse"):
        X, y, lam = gen(name)
        _, info = pq.solve_lasso(X, y, lam)
        assert info["status"] == "Solved"
        got[name] = (info["ipm_iter"], info["admm_iter"])
    assert all(1 <= v[0] <= 12 and v[1] < 1000 for v in got.values()), got


def test_lasso_rejects_bad_input(pq):
    X, y, lam = gen("wide_dense")
    _, info = pq.solve_lasso(X, y, 0.0)
    assert info["status"] == "Failure"
