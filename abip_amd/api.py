"""Host-side mirror of the reference's Matlab surface (scripts/matlab/*.m) and mex gateway
(src/abip-lp/mexfile/abip_mex.c) for the LP path, on top of libabip_hip.so.

    x, y, s, info = abip(data, K, params)          # scripts/matlab/abip.m:1-29

``data`` is a dict (or object) with sparse ``A`` and dense ``b``, ``c`` (optional warm start ``x``, ``y``,
``s``); ``K`` a dict with ``l`` (the LP cone); ``params`` a dict as returned by ``abip_get_params``.
Names, defaults, validation messages and quirks follow the reference, including the two field-name
mismatches between abip_lpsolve.m and the mex (``max_admm_iter`` vs ``max_admm_iters``,
``restart_freq`` vs ``restart_fre``) which leave those two settings at their defaults.
"""
from __future__ import annotations

import copy
import time

import numpy as np
import scipy.sparse as sp

from .solver import LINSYS_DIRECT, LINSYS_INDIRECT, Solver

__all__ = ["abip", "abip_get_params", "abip_check_params", "abip_lpsolve", "abip_direct", "abip_indirect"]


def _get(obj, key, default=None):
    if isinstance(obj, dict):
        return obj.get(key, default)
    return getattr(obj, key, default)


def _has(obj, key):
    return (key in obj) if isinstance(obj, dict) else hasattr(obj, key)


def abip_get_params() -> dict:  # scripts/matlab/abip_get_params.m:1-35
    return dict(verbose=1, normalize=1, pcg=0, max_admm_iter=1000000, max_ipm_iter=500, timelimit=3600, tol=1e-03,
                solver=-1,
                lpalg=dict(restart_thresh=100000, restart_freq=1000, feasopt=0, scaling_method=1, half_update=0),
                qcpalg=dict(rho_primal=1.0, rho_dual=1e-06, admm_tol_factor=1.0))


def abip_check_params(params: dict) -> dict:  # scripts/matlab/abip_check_params.m
    params = copy.deepcopy(params)
    d = abip_get_params()
    for k in ("verbose", "normalize", "pcg", "max_admm_iter"):
        params.setdefault(k, d[k])
    if params["max_admm_iter"] <= 0:
        raise ValueError("Invalid max_admm_iter. Must be > 0")
    params.setdefault("max_ipm_iter", d["max_ipm_iter"])
    if params["max_ipm_iter"] <= 0:
        raise ValueError("Invalid max_ipm_iter. Must be > 0")
    params.setdefault("timelimit", d["timelimit"])
    if params["timelimit"] <= 0.0:
        raise ValueError("Invalid timelimit. Must be > 0.0")
    params.setdefault("tol", d["tol"])
    if params["tol"] <= 0.0:
        raise ValueError("Invalid tol. Must be > 0.0")
    params.setdefault("solver", d["solver"])
    lp = params.setdefault("lpalg", copy.deepcopy(d["lpalg"]))
    for k, v in d["lpalg"].items():
        lp.setdefault(k, v)
    if lp["restart_thresh"] <= 0:
        raise ValueError("Invalid LP parameter restart_thresh. Must be > 0")
    if lp["restart_freq"] <= 0:
        raise ValueError("Invalid LP parameter restart_freq. Must be > 0")
    if lp["scaling_method"] not in (1, 2, 3):
        raise ValueError("Invalid scaling method. Must be one of [1, 2, 3].")
    q = params.setdefault("qcpalg", copy.deepcopy(d["qcpalg"]))
    for k, v in d["qcpalg"].items():
        q.setdefault(k, v)
    for k in ("rho_primal", "rho_dual", "admm_tol_factor"):
        if q[k] <= 0:
            raise ValueError(f"Invalid QCP parameter {k}. Must be > 0")
    return params


# settings the mex gateway reads from its second argument (abip_mex.c:183-341), with the C field they set
_MEX_FIELDS = {
    "max_ipm_iters": "max_ipm_iters", "max_admm_iters": "max_admm_iters", "eps": "eps", "cg_rate": "cg_rate",
    "alpha": "alpha", "rho_y": "rho_y", "normalize": "normalize", "scale": "scale", "sparsity_ratio": "sparsity_ratio",
    "adaptive": "adaptive", "adaptive_lookback": "adaptive_lookback", "dynamic_sigma": "dynamic_sigma",
    "dynamic_x": "dynamic_x", "dynamic_eta": "dynamic_eta", "restart_thresh": "restart_thresh", "restart_fre": "restart_fre",
    "origin_rescale": "origin_rescale", "pc_ruiz_rescale": "pc_ruiz_rescale", "qp_rescale": "qp_rescale",
    "ruiz_iter": "ruiz_iter", "hybrid_mu": "hybrid_mu", "half_update": "half_update", "avg_criterion": "avg_criterion",
    "hybrid_thresh": "hybrid_thresh", "dynamic_sigma_second": "dynamic_sigma_second", "timelimit": "max_time",
    "verbose": "verbose", "feasopt": "pfeasopt",
}


def _mex_gateway(data, settings: dict, linsys: int):
    """[x, y, s, info] = abip_direct(data, settings) / abip_indirect(data, settings)   (abip_mex.c:83-424)."""
    A = _get(data, "A")
    if A is None:
        raise ValueError("ABIPData struct must contain a matrix 'A'.")
    if not sp.issparse(A):
        raise ValueError("Input matrix A must be in sparse format.")
    b, c = _get(data, "b"), _get(data, "c")
    if b is None:
        raise ValueError("ABIPData struct must contain a vector 'b'.")
    if c is None:
        raise ValueError("ABIPData struct must contain a vector 'c'.")
    if sp.issparse(b):
        raise ValueError("Input vector b must be in dense format.")
    if sp.issparse(c):
        raise ValueError("Input vector c must be in dense format.")
    b = np.asarray(b, dtype=np.float64).ravel()
    c = np.asarray(c, dtype=np.float64).ravel()
    over = {"max_time": 3600.0, "pfeasopt": 0}
    for k, v in settings.items():
        if k in _MEX_FIELDS:        # unknown names are silently ignored, exactly like mxGetField == NULL
            over[_MEX_FIELDS[k]] = v
    warm = None
    wx, wy, wsl = _get(data, "x"), _get(data, "y"), _get(data, "s")
    if wx is not None or wy is not None or wsl is not None:  # parse_warm_start, abip_mex.c:364-366
        n, m = c.size, b.size
        warm = (np.zeros(n) if wx is None else np.asarray(wx, float).ravel(),
                np.zeros(m) if wy is None else np.asarray(wy, float).ravel(),
                np.zeros(n) if wsl is None else np.asarray(wsl, float).ravel())
    over["warm_start"] = 0 if warm is None else 1
    with Solver(A, b, c, linsys=linsys, **over) as S:
        if warm is not None:
            S.x[:], S.y[:], S.s[:] = warm
        st = S.solve()
        info = dict(status=st["status"], ipm_iter=float(st["ipm_iter"]), admm_iter=float(st["admm_iter"]), pobj=st["pobj"],
                    dobj=st["dobj"], resPri=st["res_pri"], resDual=st["res_dual"], relGap=st["rel_gap"],
                    resInfeas=st["res_infeas"], resUnbdd=st["res_unbdd"], setupTime=st["setup_time"], solveTime=st["solve_time"])
        return S.x.copy(), S.y.copy(), S.s.copy(), info


def abip_direct(data, settings):
    return _mex_gateway(data, settings, LINSYS_DIRECT)


def abip_indirect(data, settings):
    return _mex_gateway(data, settings, LINSYS_INDIRECT)


def _lpparam_convert(params: dict) -> dict:  # abipi_lpparam_convert, abip_lpsolve.m:34-61
    lp = dict(verbose=params["verbose"], normalize=params["normalize"], max_admm_iter=params["max_admm_iter"],
              max_ipm_iters=params["max_ipm_iter"], timelimit=params["timelimit"], eps=params["tol"],
              origin_rescale=0, pc_ruiz_rescale=0, qp_rescale=0)
    sm = params["lpalg"]["scaling_method"]
    if sm == 1:
        lp["pc_ruiz_rescale"] = 1
    elif sm == 2:
        lp["qp_rescale"] = 1
    elif sm == 3:
        lp["origin_rescale"] = 1
    lp["restart_thresh"] = params["lpalg"]["restart_thresh"]
    lp["restart_freq"] = params["lpalg"]["restart_freq"]
    lp["feasopt"] = params["lpalg"]["feasopt"]
    lp["half_update"] = params["lpalg"]["half_update"]
    return lp


def abip_lpsolve(data, K, params):  # scripts/matlab/abip_lpsolve.m:1-32
    if _has(K, "f") or _has(K, "q") or _has(K, "rq") or not _has(K, "l"):
        raise ValueError("Invalid conic format for LP")
    lpparams = _lpparam_convert(params)
    t0 = time.time()
    x, y, s, lpinfo = (abip_indirect if params["pcg"] else abip_direct)(data, lpparams)
    tlp = time.time() - t0
    info = dict(status=lpinfo["status"], ipm_iter=lpinfo["ipm_iter"], admm_iter=lpinfo["admm_iter"], pres=lpinfo["resPri"],
                dres=lpinfo["resDual"], gap=lpinfo["relGap"], time=tlp)
    info["pobj"] = float(np.asarray(_get(data, "c"), float).ravel() @ x)
    info["dobj"] = float(np.asarray(_get(data, "b"), float).ravel() @ y)
    info["solver"] = "abip-lp"
    return x, y, s, info


def abip(data, K, params=None):  # scripts/matlab/abip.m:1-29
    if data is None or K is None:
        raise ValueError("Invalid number of inputs. Expected at least data and K.")
    if params is None:
        params = abip_get_params()
    params = abip_check_params(params)
    if _has(K, "f") or _has(K, "q") or _has(K, "rq") or params["solver"] == 1:
        from .qcp import abip_qcpsolve
        return abip_qcpsolve(data, K, params)      # conic path (src/abip-qcp), generic QCP formulation
    return abip_lpsolve(data, K, params)
