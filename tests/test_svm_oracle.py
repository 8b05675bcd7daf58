"""CPU: the SVM-QP reformulation of the oracle (prob_type 3; restates svm_qp_config.c).  As for LASSO there is no reference
build and no stored expected value: the pin is the problem -- the returned (w, b) must minimise the soft-margin objective,
checked against scikit-learn's SVC (libsvm, an independent solver)."""
import numpy as np
import pytest

from _svm_cases import CASES, gen, hinge_objective


@pytest.fixture(scope="module")
def pq():
    from oracle import pyoracle_qcp
    pyoracle_qcp.lib()
    return pyoracle_qcp


@pytest.mark.parametrize("name", ["tall", "wide", "tall_sparse"])
def test_svmqp_reaches_the_soft_margin_minimiser(pq, name):
    from sklearn.svm import SVC
    X, y = gen(name)
    m = X.shape[0]
    lam = 1e-2
    C = 1.0 / (m * lam)
    w, b, xi, info = pq.solve_svmqp(X, y, lam, eps=1e-5, eps_p=1e-5, eps_d=1e-5, eps_g=1e-5)
    assert info["status"] == "Solved"
    sk = SVC(kernel="linear", C=C, tol=1e-10).fit(X.toarray(), y)
    ws, bs = sk.coef_.ravel(), float(sk.intercept_[0])
    f, fs = hinge_objective(X, y, C, w, b), hinge_objective(X, y, C, ws, bs)
    assert abs(f - fs) <= 1e-4 * max(1.0, abs(fs))
    assert np.max(np.abs(w - ws)) <= 5e-3 * max(1.0, np.abs(ws).max())
    assert abs(info["pobj"] - fs) <= 1e-4 * max(1.0, abs(fs))
    assert np.all(xi >= -1e-6)


@pytest.mark.parametrize("lam", [1e-2, 1e-3])
@pytest.mark.parametrize("name", ["tall", "wide", "tall_sparse", "mid"])
def test_svm_socp_reaches_the_soft_margin_minimiser(pq, name, lam):
    """prob_type 1 (svm_config.c).  The four data shapes and two lambdas walk the branches of the scale-constant table
    (svm_config.c:63-107): dm < 10 dn, dm >= 10 dn with lambda < 1, ..."""
    from sklearn.svm import SVC
    X, y = gen(name)
    m = X.shape[0]
    C = 1.0 / (m * lam)  # scripts/bench-qcp/test_svm.m:95-102
    w, b, xi, info = pq.solve_svm(X, y, C, eps=1e-5, eps_p=1e-5, eps_d=1e-5, eps_g=1e-5)
    assert info["status"] == "Solved"
    sk = SVC(kernel="linear", C=C, tol=1e-10).fit(X.toarray(), y)
    ws, bs = sk.coef_.ravel(), float(sk.intercept_[0])
    f, fs = hinge_objective(X, y, C, w, b), hinge_objective(X, y, C, ws, bs)
    # (the hinge objective of the returned point amplifies its eps-sized constraint violation by C)
    assert abs(f - fs) <= 2e-4 * max(1.0, abs(fs)) + 5e-5 * C
    assert np.max(np.abs(w - ws)) <= 1e-2 * max(1.0, np.abs(ws).max())
    assert abs(info["pobj"] - fs) <= 2e-4 * max(1.0, abs(fs)) + 5e-5 * C
