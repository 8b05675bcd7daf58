"""GPU (-m gpu): the SVM reformulations (abip_ml surface) on the device against the CPU oracle's restatement on the same inputs and
against the soft-margin minimiser itself (scikit-learn)."""
import numpy as np
import pytest

from _svm_cases import CASES, gen, hinge_objective

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
def test_svmqp_matches_the_oracle(gpu, pq, name, eps):
    """Same trajectory (materialised operator + KKT LDL' on the device, matrix-free operator + reduced Cholesky system in the oracle)."""
    X, y = gen(name)
    lam = 1e-2
    w0, b0, xi0, wi = pq.solve_svmqp(X, y, lam, eps=eps, eps_p=eps, eps_d=eps, eps_g=eps, eps_inf=eps, eps_unb=eps)
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=3, eps=eps, linsys_solver=1, verbose=0))
    assert info["status"] == wi["status"] == "Solved"
    assert info["ipm_iter"] == wi["ipm_iter"]
    assert abs(info["admm_iter"] - wi["admm_iter"]) <= max(2, wi["ipm_iter"])
    same = info["admm_iter"] == wi["admm_iter"]
    tol = 1e-7 if same else 10 * eps
    sc = max(1.0, np.abs(w0).max())
    assert np.max(np.abs(sol["w"] - w0)) <= tol * sc and abs(sol["b"] - b0) <= tol * sc and np.max(np.abs(sol["xi"] - xi0)) <= tol * max(1.0, np.abs(xi0).max())
    assert abs(info["pobj"] - wi["pobj"]) <= tol * max(1.0, abs(wi["pobj"]))


@pytest.mark.parametrize("linsys", [1, 3])
def test_svmqp_reaches_the_minimiser(gpu, linsys):
    from sklearn.svm import SVC
    X, y = gen("mid")
    m = X.shape[0]
    lam = 1e-2
    C = 1.0 / (m * lam)
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=3, eps=1e-5, linsys_solver=linsys, verbose=0))
    assert info["status"] == "Solved"
    sk = SVC(kernel="linear", C=C, tol=1e-10).fit(X.toarray(), y)
    ws, bs = sk.coef_.ravel(), float(sk.intercept_[0])
    f, fs = hinge_objective(X, y, C, sol["w"], sol["b"]), hinge_objective(X, y, C, ws, bs)
    assert abs(f - fs) <= 2e-4 * max(1.0, abs(fs))
    assert np.max(np.abs(sol["w"] - ws)) <= 1e-2 * max(1.0, np.abs(ws).max())


@pytest.mark.parametrize("eps", [1e-3, 1e-6])
@pytest.mark.parametrize("name", list(CASES))
def test_svm_socp_matches_the_oracle(gpu, pq, name, eps):
    """prob_type 1: materialised operator + KKT LDL' on the device, matrix-free operator + the block elimination of svm_config.c:725-806 in the oracle."""
    X, y = gen(name)
    C = 1.0 / (X.shape[0] * 1e-2)
    w0, b0, xi0, wi = pq.solve_svm(X, y, C, eps=eps, eps_p=eps, eps_d=eps, eps_g=eps, eps_inf=eps, eps_unb=eps)
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": C}), dict(prob_type=1, eps=eps, linsys_solver=1, verbose=0))
    assert info["status"] == wi["status"] == "Solved"
    assert info["ipm_iter"] == wi["ipm_iter"]
    assert abs(info["admm_iter"] - wi["admm_iter"]) <= max(2, wi["ipm_iter"])
    same = info["admm_iter"] == wi["admm_iter"]
    tol = 1e-7 if same else 10 * eps
    sc = max(1.0, np.abs(w0).max())
    assert np.max(np.abs(sol["w"] - w0)) <= tol * sc and abs(sol["b"] - b0) <= tol * sc and np.max(np.abs(sol["xi"] - xi0)) <= tol * max(1.0, np.abs(xi0).max())
    assert abs(info["pobj"] - wi["pobj"]) <= tol * max(1.0, abs(wi["pobj"])) and abs(info["dobj"] - wi["dobj"]) <= tol * max(1.0, abs(wi["dobj"]))
    if same:
        for k in ("res_pri", "res_dual", "gap"):
            a, b = info[k], wi["rel_gap" if k == "gap" else k]
            assert abs(a - b) <= 1e-6 * max(abs(b), eps)


@pytest.mark.parametrize("linsys", [1, 3])
def test_svm_socp_reaches_the_minimiser(gpu, linsys):
    from sklearn.svm import SVC
    X, y = gen("mid")
    C = 1.0 / (X.shape[0] * 1e-2)
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": C}), dict(prob_type=1, eps=1e-5, linsys_solver=linsys, verbose=0))
    assert info["status"] == "Solved"
    sk = SVC(kernel="linear", C=C, tol=1e-10).fit(X.toarray(), y)
    ws, bs = sk.coef_.ravel(), float(sk.intercept_[0])
    f, fs = hinge_objective(X, y, C, sol["w"], sol["b"]), hinge_objective(X, y, C, ws, bs)
    assert abs(f - fs) <= 2e-4 * max(1.0, abs(fs))
    assert np.max(np.abs(sol["w"] - ws)) <= 1e-2 * max(1.0, np.abs(ws).max())


def test_svm_socp_refuses_an_all_zero_feature(gpu):
    """svm_config.c:300-308 divides by every column norm; the reference then iterates on NaN to its iteration caps.  The device path says so at set-up;
    the QP formulation (generic scaling: norms below the floor are left alone) solves the same data."""
    import scipy.sparse as sp
    X, y = gen("tall")
    X = sp.csc_matrix(sp.hstack([X[:, :3], sp.csc_matrix((X.shape[0], 1)), X[:, 3:]]))
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": 1.0}), dict(prob_type=1, linsys_solver=1, verbose=0))
    assert info["status"] == "Failure" and info["status_val"] == -4
    sol, info = gpu.abip_ml(dict(X=X, y=y, **{"lambda": 1e-2}), dict(prob_type=3, linsys_solver=1, verbose=0))
    assert info["status"] == "Solved" and sol["w"][3] == 0.0
