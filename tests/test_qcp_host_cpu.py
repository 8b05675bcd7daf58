"""CPU: the HOST logic of the conic path -- validation, the formulation front ends (LASSO, SVM-SOCP, SVM-QP: abip_amd/csrc/qcp_formulations.h) and the scaling --
through the pure-host export abip_hip_qcp_host_probe, against the oracle's restatement of the reference (oracle/abip_qcp_oracle.c: lasso_config.c, svm_config.c,
svm_qp_config.c, qcp_config.c) stopped at the same point by its probe hook.

The oracle applies the specialised operators matrix-free, as the reference does; the product materialises each scaled operator as one sparse matrix for the device.
The test is therefore functional: A x and A' y on random vectors, the scaled b and c, and (where the formulation has them) sc_b / sc_c must agree to rounding
(1e-12 relative to the vectors' norms: the two evaluation orders differ in a few roundings per entry).  No GPU is touched."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lasso_cases
import _svm_cases
import qcp_cases

PF, PI = C.POINTER(C.c_double), C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def libs():
    import __graft_entry__ as g
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "abip_amd", "lib", "libabip_hip.so")) or not os.path.exists(os.path.join(root, "oracle", "liboracle_qcp.so")):
        g.build()
    from abip_amd import _lib
    from oracle import pyoracle_qcp as pq
    Lo, Lp = pq.lib(), _lib.load()
    Lo.orc_qcp_set_probe.restype = None
    Lo.orc_qcp_set_probe.argtypes = [C.c_int, PF, PF, PF, PF, PF, PF, PF]
    Lp.abip_hip_qcp_host_probe.restype = C.c_int
    Lp.abip_hip_qcp_host_probe.argtypes = [C.c_void_p, C.c_void_p, PF, PF, PF, PF, PF, PF, PF, PI]
    return pq, Lo, Lp


def dims(kind, dm, dn):
    if kind == 0:
        return dm + 1, 2 + 2 * dn + dm
    if kind == 1:
        return dm + dn + 1, 4 + 3 * dn + 2 * dm
    if kind == 3:
        return dm, 1 + dn + 2 * dm
    return dm, dn


def probe_both(libs, kind, A, b, c, K, lam=0.0, Q=None, seed=0, **settings):
    pq, Lo, Lp = libs
    A = sp.csc_matrix(A)
    dm, dn = A.shape
    P = pq.Problem(A, b, c if c is not None else np.zeros(dn), K, Q=Q, set_defaults=Lo.orc_qcp_set_default_settings, verbose=0, linsys_solver=1, **settings)
    P.stgs.prob_type = kind
    P.data.lambda_ = float(lam)
    if kind != 2:
        P.data.c = None
    m, n = dims(kind, dm, dn)
    rng = np.random.default_rng(100 + seed)
    x_in, y_in = rng.standard_normal(n), rng.standard_normal(m)
    f = lambda a: a.ctypes.data_as(PF)
    out = []
    # the oracle, stopped after formulation + scaling
    Ax, Aty, bs, cs, sc = np.zeros(m), np.zeros(n), np.zeros(m), np.zeros(n), np.zeros(2)
    Lo.orc_qcp_set_probe(1, f(x_in), f(y_in), f(Ax), f(Aty), f(bs), f(cs), f(sc))
    try:
        Lo.orc_qcp_solve(C.byref(P.data), C.byref(P.sol), C.byref(P.info), C.byref(P.cone))
    finally:
        Lo.orc_qcp_set_probe(0, None, None, None, None, None, None, None)
    assert P.info.status.decode() == "Probe", P.info.status
    out.append((Ax, Aty, bs, cs, sc))
    # the product's host code
    Ax2, Aty2, b2, c2, sc4, d2 = np.zeros(m), np.zeros(n), np.zeros(m), np.zeros(n), np.zeros(4), np.zeros(2, dtype=np.int32)
    rc = Lp.abip_hip_qcp_host_probe(C.addressof(P.data), C.addressof(P.cone), f(x_in), f(y_in), f(Ax2), f(Aty2), f(b2), f(c2), f(sc4), d2.ctypes.data_as(PI))
    assert rc == 0 and tuple(d2) == (m, n)
    out.append((Ax2, Aty2, b2, c2, sc4))
    return out, (x_in, y_in)


def close(a, b, tol=1e-12):
    return np.linalg.norm(a - b) <= tol * max(np.linalg.norm(b), 1e-300)


def check(out, sc=True):
    (Ax, Aty, b, c, s2), (Ax2, Aty2, b2, c2, s4) = out
    assert close(Ax2, Ax) and close(Aty2, Aty), (np.linalg.norm(Ax2 - Ax) / np.linalg.norm(Ax), np.linalg.norm(Aty2 - Aty) / np.linalg.norm(Aty))
    assert close(b2, b) and close(c2, c)
    if sc:
        assert abs(s4[0] - s2[0]) <= 1e-14 * abs(s2[0]) and abs(s4[1] - s2[1]) <= 1e-14 * abs(s2[1])


@pytest.mark.parametrize("name", ["lasso_small", "lasso_mid", "mixed", "lp"])
@pytest.mark.parametrize("variant", ["default", "ruiz_off", "pc_on", "no_scale_E", "no_scale_bc", "scale5"])
def test_generic_scaling_equals_the_oracle(libs, name, variant):
    """scale_data (qcp_config.c:26-512): origin / Ruiz / pc passes, the cone-wise averaging of E, scale_bc, `scale`."""
    data, K = qcp_cases.make(name)
    st = {"default": {}, "ruiz_off": dict(ruiz_scaling=0), "pc_on": dict(pc_scaling=1), "no_scale_E": dict(scale_E=0), "no_scale_bc": dict(scale_bc=0),
          "scale5": dict(scale=5.0)}[variant]
    out, _ = probe_both(libs, 2, data["A"], data["b"], data["c"], K, Q=data.get("Q"), **st)
    check(out)


def test_generic_with_normalize_off_scales_like_the_reference(libs):
    """scaling_qcp_data does not read `normalize` (the switch gates the un-scaling of the solution, abip.c:580-582, and one factor of the residuals, qcp_config.c:587): the data are scaled all the same."""
    data, K = qcp_cases.make("mixed")
    out, (x_in, y_in) = probe_both(libs, 2, data["A"], data["b"], data["c"], K, Q=data.get("Q"), normalize=0)
    check(out)
    assert not close(out[1][0], data["A"] @ x_in, 1e-6)


@pytest.mark.parametrize("name", sorted(_lasso_cases.CASES))
def test_lasso_front_end_equals_the_oracle(libs, name):
    """build_lasso against init_lasso + scaling_lasso_data (lasso_config.c:8-250): both sparsity branches, m < n and m > n; the materialised operator
    [e0 | 0 | D sqrt(s2) | X~ | -X~] against lasso_A_times / lasso_AT_times applied matrix-free."""
    X, y, lam = _lasso_cases.gen(name)
    m, n = X.shape
    out, _ = probe_both(libs, 0, X, y, None, {"rq": [m + 2], "l": 2 * n}, lam=lam)
    check(out, sc=False)


@pytest.mark.parametrize("name", sorted(_svm_cases.CASES))
@pytest.mark.parametrize("lam", [0.05, 1.0, 20.0])
def test_svm_socp_front_end_equals_the_oracle(libs, name, lam):
    """build_svm against init_svm + scaling_svm_data (svm_config.c:8-171, 281-391); lambda sweeps the table of scale heuristics (:63-107)."""
    X, y = _svm_cases.gen(name)
    m, n = X.shape
    if np.any(np.asarray(abs(X).sum(axis=0)).ravel() == 0):
        pytest.skip("all-zero feature column: refused by the product, divided by in the reference")
    out, _ = probe_both(libs, 1, X, y, None, {"rq": [n + 2], "l": 2 + 2 * m + 2 * n}, lam=lam)
    check(out, sc=False)


@pytest.mark.parametrize("name", sorted(_svm_cases.CASES))
def test_svm_qp_front_end_equals_the_oracle(libs, name):
    """build_svmqp against init_svmqp + scaling_svmqp_data (svm_qp_config.c)."""
    X, y = _svm_cases.gen(name)
    m, n = X.shape
    out, _ = probe_both(libs, 3, X, y, None, {"f": n + 1, "l": 2 * m}, lam=0.1)
    check(out)


def test_host_probe_rejects_what_abip_qcp_rejects(libs):
    pq, Lo, Lp = libs
    X, y, lam = _lasso_cases.gen("wide_dense")
    m, n = X.shape
    P = pq.Problem(X, y, np.zeros(n), {"rq": [m + 2], "l": 2 * n}, set_defaults=Lo.orc_qcp_set_default_settings, verbose=0)
    P.stgs.prob_type = 0
    P.data.lambda_ = 0.0                                   # LASSO needs lambda > 0
    d2 = np.zeros(2, dtype=np.int32)
    assert Lp.abip_hip_qcp_host_probe(C.addressof(P.data), C.addressof(P.cone), None, None, None, None, None, None, None, d2.ctypes.data_as(PI)) < 0
    P.data.lambda_ = lam
    P.stgs.prob_type = 7
    assert Lp.abip_hip_qcp_host_probe(C.addressof(P.data), C.addressof(P.cone), None, None, None, None, None, None, None, d2.ctypes.data_as(PI)) < 0
