"""ctypes binding of the conic entry point abip_qcp() (include/abip_qcp.h) and the mirror of the reference's
`abip_qcp` mex call / abip_qcpsolve.m for the generic QCP formulation."""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import scipy.sparse as sp

from . import _lib

ci, cf = C.c_int, C.c_double
PF, PI = C.POINTER(cf), C.POINTER(ci)


class QCPMatrix(C.Structure):
    _fields_ = [("x", PF), ("i", PI), ("p", PI), ("m", ci), ("n", ci)]


class QCPCone(C.Structure):  # src/abip-qcp/include/abip.h:67-76
    _fields_ = [("q", PI), ("qsize", ci), ("rq", PI), ("rqsize", ci), ("f", ci), ("z", ci), ("l", ci)]


class QCPSettings(C.Structure):  # src/abip-qcp/include/abip.h:93-131
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


def _bind():
    L = _lib.load()
    if not getattr(L, "_qcp_bound", False):
        L.abip_qcp.restype = ci
        L.abip_qcp.argtypes = [C.POINTER(QCPData), C.POINTER(QCPSolution), C.POINTER(QCPInfo), C.POINTER(QCPCone)]
        L.abip_qcp_set_default_settings.restype = None
        L.abip_qcp_set_default_settings.argtypes = [C.POINTER(QCPData)]
        L._qcp_bound = True
    return L


def _csc(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    x = np.array(M.data, dtype=np.float64, copy=True)
    i = np.array(M.indices, dtype=np.int32, copy=True)
    p = np.array(M.indptr, dtype=np.int32, copy=True)
    return (x, i, p), QCPMatrix(x.ctypes.data_as(PF), i.ctypes.data_as(PI), p.ctypes.data_as(PI), M.shape[0], M.shape[1])


def _get(obj, key, default=None):
    return obj.get(key, default) if isinstance(obj, dict) else getattr(obj, key, default)


def abip_qcp(data, cones, settings: dict):
    """[sol, info] = abip_qcp(data, cones, settings)   (src/abip-qcp/mex/abip_qcp_mex.c:109-525)."""
    L = _bind()
    A, b, c, Q = _get(data, "A"), _get(data, "b"), _get(data, "c"), _get(data, "Q")
    if c is None:
        raise ValueError("ABIPData struct must contain a `c` entry.")
    if A is None or b is None:
        raise ValueError("the device path needs `A` and `b`")
    if not sp.issparse(A):
        raise ValueError("Input matrix A must be in sparse format (pass in sparse(A))")
    if Q is not None and not sp.issparse(Q):
        raise ValueError("Input matrix Q must be in sparse format (pass in sparse(Q))")
    keep = []
    (ka, Am) = _csc(A); keep.append(ka)
    Qm = None
    if Q is not None:
        (kq, Qm) = _csc(Q); keep.append(kq)
    m, n = Am.m, Am.n
    b = np.array(b, dtype=np.float64, copy=True).ravel(); c = np.array(c, dtype=np.float64, copy=True).ravel()
    stgs = QCPSettings()
    d = QCPData(m, n, C.pointer(Am), C.pointer(Qm) if Qm is not None else None, b.ctypes.data_as(PF), c.ctypes.data_as(PF), 0.0, C.pointer(stgs))
    L.abip_qcp_set_default_settings(C.byref(d))
    if "eps" in settings:                       # abip_qcp_mex.c:307-314
        for k in ("eps", "eps_p", "eps_d", "eps_g", "eps_inf", "eps_unb"):
            setattr(stgs, k, settings["eps"])
    for k, v in settings.items():               # unknown names are ignored like mxGetField == NULL
        if k != "eps" and hasattr(stgs, k) and k != "prob_type":
            setattr(stgs, k, int(v) if isinstance(getattr(stgs, k), int) else float(v))
    stgs.prob_type = 2                          # abip_qcp_mex.c:436
    q = np.array(_get(cones, "q", []) or [], dtype=np.int32).ravel(); rq = np.array(_get(cones, "rq", []) or [], dtype=np.int32).ravel()
    K = QCPCone(q.ctypes.data_as(PI) if q.size else None, int(q.size), rq.ctypes.data_as(PI) if rq.size else None, int(rq.size),
                int(_get(cones, "f", 0) or 0), int(_get(cones, "z", 0) or 0), int(_get(cones, "l", 0) or 0))
    x = np.full(n, np.nan); y = np.full(m, np.nan); s = np.full(n, np.nan)
    sol = QCPSolution(x.ctypes.data_as(PF), y.ctypes.data_as(PF), s.ctypes.data_as(PF))
    info = QCPInfo()
    L.abip_qcp(C.byref(d), C.byref(sol), C.byref(info), C.byref(K))
    out = dict(ipm_iter=info.ipm_iter, admm_iter=info.admm_iter, status=info.status.decode(), pobj=info.pobj, dobj=info.dobj,
               res_pri=info.res_pri, res_dual=info.res_dual, gap=info.rel_gap, status_val=info.status_val,
               setup_time=info.setup_time / 1e3, solve_time=info.solve_time / 1e3, runtime=(info.setup_time + info.solve_time) / 1e3,
               lin_sys_time_per_iter=info.avg_linsys_time / 1e3, avg_cg_iters=info.avg_cg_iters)
    st = (C.c_double * 8)()
    L.abip_hip_qcp_last_stats(st)
    out["factor"] = dict(N=int(st[0]), dense_tail=int(st[1]), lnnz=int(st[2]), levels=[int(st[3]), int(st[4])], solves_timed=int(st[5]),
                         solve_ms_total=float(st[6]), head_nnz=int(st[7]))
    ph = (C.c_double * 5)()
    L.abip_hip_qcp_phase_times(ph)   # the reference's per-phase timers (abip.c:1084-1093), seconds
    out["phase_times"] = dict(project_lin_sys=ph[0], solve_barrier_subproblem=ph[1], calc_residuals=ph[2], err_inner=ph[3], updating_work=ph[4])
    return dict(x=x, y=y, s=s), out


def abip_ml(data, settings: dict):
    """[sol, info] = abip_ml(data, settings)   (src/abip-qcp/mex/abip_ml_mex.c:90-449): the machine-learning front end.
    data: X (sparse), y (dense), lambda; settings.prob_type is mandatory (:266-276).  Served: prob_type 0 (LASSO,
    min 1/2 |X beta - y|^2 + lambda |beta|_1; sol = {x: beta}), prob_type 1 (soft-margin SVM as an SOCP, min 1/2 |w|^2 + lambda sum xi;
    labels in y) and prob_type 3 (the same SVM as a QP with C = 1 / (m lambda)).  The gateway builds the cone itself (:315-342)."""
    L = _bind()
    X, y, lam = _get(data, "X"), _get(data, "y"), _get(data, "lambda")
    if X is None:
        raise ValueError("ABIPData struct must contain a `X` entry.")
    if not sp.issparse(X):
        raise ValueError("Input matrix X must be in sparse format (pass in sparse(X))")
    if y is None:
        raise ValueError("ABIPData struct must contain a `y` entry.")
    if sp.issparse(y):
        raise ValueError("Input vector y must be in dense format (pass in full(y))")
    if lam is None:
        raise ValueError("ABIPData struct must contain a `lambda` entry.")
    if "prob_type" not in settings:
        raise ValueError("Please input the machine learning problem type")
    prob_type = int(settings["prob_type"])
    if prob_type not in (0, 1, 3):
        raise ValueError("Invalid problem type")
    (keep, Xm) = _csc(X)
    m, n = Xm.m, Xm.n
    y = np.array(y, dtype=np.float64, copy=True).ravel()
    stgs = QCPSettings()
    d = QCPData(m, n, C.pointer(Xm), None, y.ctypes.data_as(PF), None, float(lam), C.pointer(stgs))
    L.abip_qcp_set_default_settings(C.byref(d))
    if "eps" in settings:                       # abip_ml_mex.c:176-184
        for k in ("eps", "eps_p", "eps_d", "eps_g", "eps_inf", "eps_unb"):
            setattr(stgs, k, settings["eps"])
    for k, v in settings.items():
        if k != "eps" and hasattr(stgs, k):
            setattr(stgs, k, int(v) if isinstance(getattr(stgs, k), int) else float(v))
    stgs.prob_type = prob_type
    rq = np.array([2 + m], dtype=np.int32)
    beta = np.full(n, np.nan); b0 = np.full(1, np.nan); xi = np.full(m, np.nan)
    if prob_type == 0:                          # :328-331
        K = QCPCone(None, 0, rq.ctypes.data_as(PI), 1, 0, 0, 2 * n)
        sol = QCPSolution(beta.ctypes.data_as(PF), None, None)
    elif prob_type == 1:                        # SVM-SOCP, :333-336
        rq[0] = 2 + n
        K = QCPCone(None, 0, rq.ctypes.data_as(PI), 1, 0, 0, 2 + 2 * m + 2 * n)
        sol = QCPSolution(beta.ctypes.data_as(PF), b0.ctypes.data_as(PF), xi.ctypes.data_as(PF))
    else:                                       # SVM-QP, :338-342
        K = QCPCone(None, 0, None, 0, n + 1, 0, 2 * m)
        sol = QCPSolution(beta.ctypes.data_as(PF), b0.ctypes.data_as(PF), xi.ctypes.data_as(PF))
    info = QCPInfo()
    L.abip_qcp(C.byref(d), C.byref(sol), C.byref(info), C.byref(K))
    out = dict(ipm_iter=info.ipm_iter, admm_iter=info.admm_iter, status=info.status.decode(), pobj=info.pobj, dobj=info.dobj,
               res_pri=info.res_pri, res_dual=info.res_dual, gap=info.rel_gap, status_val=info.status_val,
               setup_time=info.setup_time / 1e3, solve_time=info.solve_time / 1e3, runtime=(info.setup_time + info.solve_time) / 1e3,
               lin_sys_time_per_iter=info.avg_linsys_time / 1e3, avg_cg_iters=info.avg_cg_iters)
    st = (C.c_double * 8)()
    L.abip_hip_qcp_last_stats(st)
    out["factor"] = dict(N=int(st[0]), dense_tail=int(st[1]), lnnz=int(st[2]), levels=[int(st[3]), int(st[4])], solves_timed=int(st[5]),
                         solve_ms_total=float(st[6]), head_nnz=int(st[7]))
    if prob_type == 0:
        return dict(x=beta), out
    # (the gateway's own output switch tests prob_type against 2 and 4, abip_ml_mex.c:362, so Matlab receives {x: w} for SVM too;
    #  the C entry point hands back all three, un_scaling_svmqp_sol svm_qp_config.c:595-619)
    return dict(x=beta, w=beta, b=float(b0[0]), xi=xi), out


def partition_columns(A, cones, world: int) -> np.ndarray:
    """Column bounds of the sharded conic path (abip_hip_qcp_dist_partition; pure host code, also what abip_qcp() uses): world + 1 entries."""
    L = _bind()
    L.abip_hip_qcp_dist_partition.restype = ci
    L.abip_hip_qcp_dist_partition.argtypes = [C.POINTER(QCPMatrix), C.POINTER(QCPCone), ci, PI]
    (keep, Am) = _csc(A)
    q = np.array(_get(cones, "q", []) or [], dtype=np.int32).ravel(); rq = np.array(_get(cones, "rq", []) or [], dtype=np.int32).ravel()
    K = QCPCone(q.ctypes.data_as(PI) if q.size else None, int(q.size), rq.ctypes.data_as(PI) if rq.size else None, int(rq.size),
                int(_get(cones, "f", 0) or 0), int(_get(cones, "z", 0) or 0), int(_get(cones, "l", 0) or 0))
    out = np.zeros(world + 1, dtype=np.int32)
    rc = L.abip_hip_qcp_dist_partition(C.byref(Am), C.byref(K), int(world), out.ctypes.data_as(PI))
    if rc != 0:
        raise ValueError(f"cannot partition the columns over {world} ranks ({rc})")
    return out


def cone_prox(kind: int, tmp, lam: float, x_prev=None):
    """One SOC (kind 0) / rotated-SOC (kind 1) barrier prox on the device (kq_cones); unit-level mirror of cones.c:130-248."""
    L = _bind()
    t = np.ascontiguousarray(tmp, dtype=np.float64)
    x = np.zeros_like(t) if x_prev is None else np.array(x_prev, dtype=np.float64, copy=True)
    L.abip_hip_qcp_cone_prox.restype = ci
    L.abip_hip_qcp_cone_prox.argtypes = [ci, PF, PF, C.c_double, ci]
    rc = L.abip_hip_qcp_cone_prox(int(kind), x.ctypes.data_as(PF), t.ctypes.data_as(PF), float(lam), int(t.size))
    if rc != 0:
        raise RuntimeError(f"abip_hip_qcp_cone_prox failed ({rc})")
    return x


def abip_qcpsolve(data, K, params):
    """scripts/matlab/abip_qcpsolve.m:1-24 (+ abipi_qcpparam_convert :26-52, abipi_qcpinfo_convert :54-66)."""
    q = dict(verbose=params["verbose"], normalize=params["normalize"], max_admm_iter=params["max_admm_iter"],
             max_ipm_iters=params["max_ipm_iter"], timelimit=params["timelimit"], linsys_solver=3 if params["pcg"] else 1,
             eps_p=params["tol"], eps_d=params["tol"], eps_g=params["tol"], rho_x=params["qcpalg"]["rho_primal"],
             rho_y=params["qcpalg"]["rho_dual"], psi=params["qcpalg"]["admm_tol_factor"])
    t0 = time.time()
    sol, qi = abip_qcp(data, K, q)
    info = dict(status=qi["status"], ipm_iter=qi["ipm_iter"], admm_iter=qi["admm_iter"], pres=qi["res_pri"], dres=qi["res_dual"],
                gap=qi["gap"], pobj=qi["pobj"], dobj=qi["dobj"], time=time.time() - t0, solver="abip-qcp")
    return sol["x"], sol["y"], sol["s"], info
