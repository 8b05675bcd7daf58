"""ctypes access to the QCP CPU oracle (oracle/liboracle_qcp.so) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ci, cf = C.c_int, C.c_double
PF, PI = C.POINTER(cf), C.POINTER(ci)


class QCPMatrix(C.Structure):
    _fields_ = [("x", PF), ("i", PI), ("p", PI), ("m", ci), ("n", ci)]


class QCPCone(C.Structure):
    _fields_ = [("q", PI), ("qsize", ci), ("rq", PI), ("rqsize", ci), ("f", ci), ("z", ci), ("l", ci)]


class QCPSettings(C.Structure):
    _fields_ = [("normalize", ci), ("scale_E", ci), ("scale_bc", ci), ("scale", cf), ("rho_x", cf), ("rho_y", cf), ("rho_tau", cf),
                ("max_ipm_iters", ci), ("max_admm_iters", ci), ("eps", cf), ("eps_p", cf), ("eps_d", cf), ("eps_g", cf), ("eps_inf", cf),
                ("eps_unb", cf), ("err_dif", cf), ("alpha", cf), ("cg_rate", cf), ("use_indirect", ci), ("inner_check_period", ci),
                ("outer_check_period", ci), ("verbose", ci), ("linsys_solver", ci), ("prob_type", ci), ("time_limit", cf), ("psi", cf),
                ("origin_scaling", ci), ("ruiz_scaling", ci), ("pc_scaling", ci)]


class QCPData(C.Structure):
    _fields_ = [("m", ci), ("n", ci), ("A", C.POINTER(QCPMatrix)), ("Q", C.POINTER(QCPMatrix)), ("b", PF), ("c", PF), ("lambda_", cf),
                ("stgs", C.POINTER(QCPSettings))]


class QCPSolution(C.Structure):
    _fields_ = [("x", PF), ("y", PF), ("s", PF)]


class QCPInfo(C.Structure):
    _fields_ = [("status", C.c_char * 32), ("status_val", ci), ("ipm_iter", ci), ("admm_iter", ci), ("pobj", cf), ("dobj", cf),
                ("res_pri", cf), ("res_dual", cf), ("rel_gap", cf), ("res_infeas", cf), ("res_unbdd", cf), ("setup_time", cf),
                ("solve_time", cf), ("avg_linsys_time", cf), ("avg_cg_iters", cf)]


def _csc(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    x = np.array(M.data, dtype=np.float64, copy=True)
    i = np.array(M.indices, dtype=np.int32, copy=True)
    p = np.array(M.indptr, dtype=np.int32, copy=True)
    return (x, i, p), QCPMatrix(x.ctypes.data_as(PF), i.ctypes.data_as(PI), p.ctypes.data_as(PI), M.shape[0], M.shape[1])


class Problem:
    """Owns the ctypes views of one conic problem (shared by the oracle wrapper and the product's Python mirror tests)."""

    def __init__(self, A, b, c, K: dict, Q=None, set_defaults=None, **settings):
        self.keep = []
        (ka, self.A) = _csc(A); self.keep.append(ka)
        self.m, self.n = self.A.m, self.A.n
        self.Q = None
        if Q is not None:
            (kq, self.Q) = _csc(Q); self.keep.append(kq)
        self.b = np.array(b, dtype=np.float64, copy=True); self.c = np.array(c, dtype=np.float64, copy=True)
        self.stgs = QCPSettings()
        self.data = QCPData(self.m, self.n, C.pointer(self.A), C.pointer(self.Q) if self.Q is not None else None,
                            self.b.ctypes.data_as(PF), self.c.ctypes.data_as(PF), 0.0, C.pointer(self.stgs))
        set_defaults(C.byref(self.data))
        for k, v in settings.items():
            if not hasattr(self.stgs, k):
                raise KeyError(k)
            setattr(self.stgs, k, v)
        q = np.array(K.get("q", []), dtype=np.int32).ravel(); rq = np.array(K.get("rq", []), dtype=np.int32).ravel()
        self.keep += [q, rq]
        self.cone = QCPCone(q.ctypes.data_as(PI) if q.size else None, int(q.size), rq.ctypes.data_as(PI) if rq.size else None, int(rq.size),
                            int(K.get("f", 0)), int(K.get("z", 0)), int(K.get("l", 0)))
        self.x = np.full(self.n, np.nan); self.y = np.full(self.m, np.nan); self.s = np.full(self.n, np.nan)
        self.sol = QCPSolution(self.x.ctypes.data_as(PF), self.y.ctypes.data_as(PF), self.s.ctypes.data_as(PF))
        self.info = QCPInfo()

    def info_dict(self):
        d = {k: getattr(self.info, k) for k, _ in QCPInfo._fields_ if k != "status"}
        d["status"] = self.info.status.decode()
        return d


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle_qcp.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", HERE, "oracle"], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.orc_qcp_solve.restype = ci
        L.orc_qcp_solve.argtypes = [C.POINTER(QCPData), C.POINTER(QCPSolution), C.POINTER(QCPInfo), C.POINTER(QCPCone)]
        L.orc_qcp_set_default_settings.argtypes = [C.POINTER(QCPData)]
        L.orc_qcp_set_trace.argtypes = [ci, PF]
        L.orc_qcp_trace_count.restype = ci
        L.orc_qcp_cone_prox.restype = None
        L.orc_qcp_cone_prox.argtypes = [ci, PF, PF, cf, ci]
        _lib = L
    return _lib


def solve(A, b, c, K, Q=None, trace: int = 0, **settings):
    L = lib()
    P = Problem(A, b, c, K, Q=Q, set_defaults=L.orc_qcp_set_default_settings, verbose=0, **settings)
    tr = None
    if trace:
        tr = np.zeros((trace, 3, P.m + P.n + 1))
        L.orc_qcp_set_trace(trace, tr.ctypes.data_as(PF))
    L.orc_qcp_solve(C.byref(P.data), C.byref(P.sol), C.byref(P.info), C.byref(P.cone))
    if trace:
        tr = tr[: L.orc_qcp_trace_count()]
        L.orc_qcp_set_trace(0, None)
    return P.x.copy(), P.y.copy(), P.s.copy(), P.info_dict(), tr


def solve_lasso(X, y, lam: float, **settings):
    """min 1/2 |X beta - y|^2 + lam |beta|_1 through the LASSO reformulation (prob_type 0, lasso_config.c); returns (beta, info).
    The caller hands over what abip_ml_mex.c:117-160,322-331 does: X, y, lambda, the cone {rq: [m + 2], l: 2 n}, no c."""
    L = lib()
    X = sp.csc_matrix(X)
    m, n = X.shape
    settings.setdefault("linsys_solver", 1)  # (the default rule of util.c:237-243 picks 5, dense Cholesky, for dense data)
    P = Problem(X, y, np.zeros(n), {"rq": [m + 2], "l": 2 * n}, set_defaults=L.orc_qcp_set_default_settings, verbose=0, **settings)
    P.stgs.prob_type = 0
    P.data.lambda_ = float(lam)
    P.data.c = None
    q, p_ = 2 + 2 * n + m, m + 1
    P.x = np.full(q, np.nan); P.y = np.full(p_, np.nan); P.s = np.full(q, np.nan)
    P.sol = QCPSolution(P.x.ctypes.data_as(PF), P.y.ctypes.data_as(PF), P.s.ctypes.data_as(PF))
    L.orc_qcp_solve(C.byref(P.data), C.byref(P.sol), C.byref(P.info), C.byref(P.cone))
    return P.x[:n].copy(), P.info_dict()


def solve_svmqp(X, y, lam: float, **settings):
    """Soft-margin SVM min 1/2 |w|^2 + 1/(m lam) sum xi, y_i (x_i'w + b) >= 1 - xi_i, through the QP reformulation
    (prob_type 3, svm_qp_config.c); returns (w, b, xi, info).  Cone as abip_ml_mex.c:338-342: f = n + 1, l = 2 m."""
    L = lib()
    X = sp.csc_matrix(X)
    m, n = X.shape
    settings.setdefault("linsys_solver", 1)
    P = Problem(X, y, np.zeros(n), {"f": n + 1, "l": 2 * m}, set_defaults=L.orc_qcp_set_default_settings, verbose=0, **settings)
    P.stgs.prob_type = 3
    P.data.lambda_ = float(lam)
    P.data.c = None
    q = 1 + n + 2 * m
    P.x = np.full(q, np.nan); P.y = np.full(m, np.nan); P.s = np.full(q, np.nan)
    P.sol = QCPSolution(P.x.ctypes.data_as(PF), P.y.ctypes.data_as(PF), P.s.ctypes.data_as(PF))
    L.orc_qcp_solve(C.byref(P.data), C.byref(P.sol), C.byref(P.info), C.byref(P.cone))
    return P.x[:n].copy(), float(P.y[0]), P.s[:m].copy(), P.info_dict()


def solve_svm(X, y, C_: float, **settings):
    """Soft-margin SVM min 1/2 |w|^2 + C sum xi through the SOCP reformulation (prob_type 1, svm_config.c; data.lambda = C,
    scripts/bench-qcp/test_svm.m:95-102); returns (w, b, xi, info).  Cone as abip_ml_mex.c:333-336: rq = [n + 2], l = 2 + 2 m + 2 n."""
    L = lib()
    X = sp.csc_matrix(X)
    m, n = X.shape
    settings.setdefault("linsys_solver", 1)
    P = Problem(X, y, np.zeros(n), {"rq": [n + 2], "l": 2 + 2 * m + 2 * n}, set_defaults=L.orc_qcp_set_default_settings, verbose=0, **settings)
    P.stgs.prob_type = 1
    P.data.lambda_ = float(C_)
    P.data.c = None
    q, p_ = 4 + 3 * n + 2 * m, m + n + 1
    P.x = np.full(q, np.nan); P.y = np.full(p_, np.nan); P.s = np.full(q, np.nan)
    P.sol = QCPSolution(P.x.ctypes.data_as(PF), P.y.ctypes.data_as(PF), P.s.ctypes.data_as(PF))
    L.orc_qcp_solve(C.byref(P.data), C.byref(P.sol), C.byref(P.info), C.byref(P.cone))
    return P.x[:n].copy(), float(P.y[0]), P.s[:m].copy(), P.info_dict()


def cone_prox(kind: int, tmp, lam: float, x_prev=None):
    """Barrier prox of one cone (cones.c:130-288): kind 0 SOC, 1 rotated SOC, 2 orthant."""
    L = lib()
    t = np.ascontiguousarray(tmp, dtype=np.float64)
    x = np.zeros_like(t) if x_prev is None else np.array(x_prev, dtype=np.float64, copy=True)
    L.orc_qcp_cone_prox(int(kind), x.ctypes.data_as(PF), t.ctypes.data_as(PF), float(lam), int(t.size))
    return x
