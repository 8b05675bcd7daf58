"""ctypes access to the CPU checkers -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Two checkers share one calling convention (the reference's struct layouts,
include/abip.h):

* ``ref``    -- the REAL reference compiled by oracle/Makefile into
  oracle/_ref/libabip_ref_{direct,indirect}.so (only where /root/reference was
  available at build time; the prebuilt .so files travel to the GPU box).
* ``oracle`` -- our own plain-C restatement, oracle/liboracle_lp.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
c_int = C.c_long      # abip_int with DLONG
c_flt = C.c_double
PF = C.POINTER(c_flt)
PI = C.POINTER(c_int)


class ABIPMatrix(C.Structure):
    _fields_ = [("x", PF), ("i", PI), ("p", PI), ("m", c_int), ("n", c_int)]


class ABIPSettings(C.Structure):
    _fields_ = [
        ("normalize", c_int), ("pfeasopt", c_int), ("scale", c_flt), ("rho_y", c_flt), ("sparsity_ratio", c_flt),
        ("max_ipm_iters", c_int), ("max_admm_iters", c_int), ("max_time", c_flt),
        ("eps", c_flt), ("alpha", c_flt), ("cg_rate", c_flt),
        ("adaptive", c_int), ("eps_cor", c_flt), ("eps_pen", c_flt),
        ("dynamic_sigma", c_flt), ("dynamic_x", c_flt), ("dynamic_eta", c_flt),
        ("restart_fre", c_int), ("restart_thresh", c_int),
        ("verbose", c_int), ("warm_start", c_int), ("adaptive_lookback", c_int),
        ("origin_rescale", c_int), ("pc_ruiz_rescale", c_int), ("qp_rescale", c_int), ("ruiz_iter", c_int),
        ("hybrid_mu", c_int), ("hybrid_thresh", c_flt), ("dynamic_sigma_second", c_flt),
        ("half_update", c_int), ("avg_criterion", c_int),
    ]


class ABIPData(C.Structure):
    _fields_ = [("m", c_int), ("n", c_int), ("A", C.POINTER(ABIPMatrix)), ("b", PF), ("c", PF), ("sp", c_flt),
                ("stgs", C.POINTER(ABIPSettings))]


class ABIPSolution(C.Structure):
    _fields_ = [("x", PF), ("y", PF), ("s", PF)]


class ABIPInfo(C.Structure):
    _fields_ = [("status", C.c_char * 32), ("status_val", c_int), ("ipm_iter", c_int), ("admm_iter", c_int),
                ("pobj", c_flt), ("dobj", c_flt), ("res_pri", c_flt), ("res_dual", c_flt), ("rel_gap", c_flt),
                ("res_infeas", c_flt), ("res_unbdd", c_flt), ("setup_time", c_flt), ("solve_time", c_flt)]


class RefWork(C.Structure):
    """struct ABIP_WORK of the reference (src/abip-lp/include/abip.h:126-176), DLONG build."""
    _fields_ = [("sigma", c_flt), ("gamma", c_flt), ("final_check", c_int), ("double_check", c_int),
                ("mu", c_flt), ("beta", c_flt),
                ("u", PF), ("v", PF), ("u_t", PF), ("u_prev", PF), ("v_prev", PF), ("u_avg", PF), ("v_avg", PF),
                ("u_avgcon", PF), ("v_avgcon", PF), ("u_sumcon", PF), ("v_sumcon", PF), ("fre_old", c_int),
                ("h", PF), ("g", PF), ("pr", PF), ("dr", PF),
                ("g_th", c_flt), ("sc_b", c_flt), ("sc_c", c_flt), ("nm_b", c_flt), ("nm_c", c_flt),
                ("b", PF), ("c", PF), ("m", c_int), ("n", c_int), ("A", C.POINTER(ABIPMatrix)), ("sp", c_flt),
                ("p", C.c_void_p), ("adapt", C.c_void_p), ("stgs", C.POINTER(ABIPSettings)), ("scal", C.c_void_p)]


DEFAULTS = dict(  # src/abip-lp/src/util.c:288-329 + the two fields only the mex sets (abip_mex.c:320-341)
    max_ipm_iters=500, max_admm_iters=1000000, eps=1e-3, alpha=1.8, cg_rate=2.0, normalize=1, scale=1.0,
    rho_y=1e-3, sparsity_ratio=0.01, adaptive=1, eps_cor=0.2, eps_pen=0.1, adaptive_lookback=20,
    dynamic_x=0.8, dynamic_eta=1.1, restart_fre=1000, restart_thresh=100000, origin_rescale=0,
    pc_ruiz_rescale=1, qp_rescale=0, ruiz_iter=10, hybrid_mu=1, dynamic_sigma=-1.0, hybrid_thresh=1000.0,
    dynamic_sigma_second=0.5, half_update=0, avg_criterion=0, verbose=0, warm_start=0, max_time=3600.0, pfeasopt=0,
)


def make_settings(**over) -> ABIPSettings:
    s = ABIPSettings()
    vals = dict(DEFAULTS)
    for k, v in over.items():
        if k not in vals:
            raise KeyError(f"unknown ABIP setting {k!r}")
        vals[k] = v
    for k, v in vals.items():
        setattr(s, k, v)
    return s


def build(ref: bool = True, quiet: bool = True) -> None:
    """(Re)build the checkers.  Building is allowed anywhere; *using* them only from tests/bench/smoke."""
    args = ["make", "-C", HERE, "oracle"] + (["ref"] if ref else [])
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL if quiet else None)


_libs: dict = {}


def lib(kind: str, linsys: str = "indirect"):
    """kind in {'oracle','oracle_omp','ref'}; linsys in {'direct','indirect'} (only meaningful for 'ref')."""
    key = (kind, linsys if kind == "ref" else "")
    if key in _libs:
        return _libs[key]
    if kind in ("oracle", "oracle_omp"):
        path = os.path.join(HERE, "liboracle_lp.so" if kind == "oracle" else "liboracle_lp_omp.so")
        if not os.path.exists(path):
            build(ref=False)
    else:
        path = os.path.join(HERE, "_ref", f"libabip_ref_{linsys}.so")
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    L = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0))
    if kind in ("oracle", "oracle_omp"):
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_get_threads.restype = C.c_int
        L.orc_lp_init.restype = C.c_void_p
        L.orc_lp_init.argtypes = [C.POINTER(ABIPData), C.POINTER(ABIPInfo), C.c_int]
        L.orc_lp_solve.restype = c_int
        L.orc_lp_solve.argtypes = [C.c_void_p, C.POINTER(ABIPData), C.POINTER(ABIPSolution), C.POINTER(ABIPInfo)]
        L.orc_lp_finish.argtypes = [C.c_void_p]
        L.orc_lp_vec.restype = PF
        L.orc_lp_vec.argtypes = [C.c_void_p, C.c_char_p, PI]
        L.orc_lp_scalar.restype = c_flt
        L.orc_lp_scalar.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_lp_set_trace.argtypes = [C.c_void_p, c_int, PF]
        L.orc_lp_trace_count.restype = c_int
        L.orc_lp_trace_count.argtypes = [C.c_void_p]
        L.orc_lp_kkt_solve.restype = c_int
        L.orc_lp_kkt_solve.argtypes = [C.c_void_p, PF, PF, c_int]
        L.orc_accum_by_Atrans.argtypes = [c_int, PF, PI, PI, PF, PF]
        L.orc_accum_by_A.argtypes = [c_int, PF, PI, PI, PF, PF]
        L.orc_normalize_A.argtypes = [C.POINTER(ABIPMatrix), C.POINTER(ABIPSettings), PF, PF, PF, PF]
    else:
        L.abip_init.restype = C.POINTER(RefWork)
        L.abip_init.argtypes = [C.POINTER(ABIPData), C.POINTER(ABIPInfo)]
        L.abip_solve.restype = c_int
        L.abip_solve.argtypes = [C.POINTER(RefWork), C.POINTER(ABIPData), C.POINTER(ABIPSolution), C.POINTER(ABIPInfo)]
        L.abip_finish.argtypes = [C.POINTER(RefWork)]
    _libs[key] = L
    return L


def have_ref() -> bool:
    return all(os.path.exists(os.path.join(HERE, "_ref", f"libabip_ref_{k}.so")) for k in ("direct", "indirect"))


def _f(a):
    return a.ctypes.data_as(PF)


def _i(a):
    return a.ctypes.data_as(PI)


@dataclass
class Result:
    x: np.ndarray
    y: np.ndarray
    s: np.ndarray
    info: dict
    work: dict = field(default_factory=dict)   # copies of u, v, u_t, scalars after the solve
    trace: np.ndarray | None = None            # (T, 3, l) for the oracle when requested
    settings_after: dict = field(default_factory=dict)


def _info_dict(info: ABIPInfo) -> dict:
    d = {k: getattr(info, k) for k, _ in ABIPInfo._fields_ if k != "status"}
    d["status"] = info.status.decode()
    return d


class Problem:
    """Owns contiguous copies of the CSC arrays (the reference scales A in place)."""

    def __init__(self, A, b, c, **settings):
        import scipy.sparse as sp
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.m, self.n = A.shape
        self.Ax = np.array(A.data, dtype=np.float64, copy=True)
        self.Ai = np.array(A.indices, dtype=np.int64, copy=True)
        self.Ap = np.array(A.indptr, dtype=np.int64, copy=True)
        self.b = np.array(b, dtype=np.float64, copy=True)
        self.c = np.array(c, dtype=np.float64, copy=True)
        self.stgs = make_settings(**settings)
        self.mat = ABIPMatrix(_f(self.Ax), _i(self.Ai), _i(self.Ap), self.m, self.n)
        self.data = ABIPData(self.m, self.n, C.pointer(self.mat), _f(self.b), _f(self.c),
                             float(self.Ax.size) / (float(self.m) * float(self.n)), C.pointer(self.stgs))


def solve(kind: str, A, b, c, linsys: str = "indirect", trace: int = 0, warm=None, **settings) -> Result:
    """Run a full solve on the chosen checker.  ``kind`` = 'ref' | 'oracle'."""
    P = Problem(A, b, c, **settings)
    L = lib(kind, linsys)
    info = ABIPInfo()
    m, n = P.m, P.n
    x = np.full(n, np.nan); y = np.full(m, np.nan); s = np.full(n, np.nan)
    if warm is not None:
        x[:] = warm[0]; y[:] = warm[1]; s[:] = warm[2]
    sol = ABIPSolution(_f(x), _f(y), _f(s))
    l = m + n + 1
    tr = None
    work = {}
    if kind in ("oracle", "oracle_omp"):
        w = L.orc_lp_init(C.byref(P.data), C.byref(info), 0 if linsys == "direct" else 1)
        if not w:
            return Result(x, y, s, dict(status="Failure", status_val=-4))
        if trace:
            tr = np.zeros((trace, 3, l))
            L.orc_lp_set_trace(w, trace, _f(tr))
        L.orc_lp_solve(w, C.byref(P.data), C.byref(sol), C.byref(info))
        if trace:
            tr = tr[: L.orc_lp_trace_count(w)]
        for nm in ("u", "v", "u_t", "g", "h", "D", "E", "b", "c", "u_avgcon", "v_avgcon"):
            ln = c_int(0)
            p = L.orc_lp_vec(w, nm.encode(), C.byref(ln))
            work[nm] = np.ctypeslib.as_array(p, shape=(ln.value,)).copy()
        for nm in ("mu", "beta", "sigma", "gamma", "g_th", "sc_b", "sc_c", "tot_cg_its", "lnnz"):
            work[nm] = L.orc_lp_scalar(w, nm.encode())
        L.orc_lp_finish(w)
    else:
        w = L.abip_init(C.byref(P.data), C.byref(info))
        if not w:
            return Result(x, y, s, dict(status="Failure", status_val=-4))
        L.abip_solve(w, C.byref(P.data), C.byref(sol), C.byref(info))
        W = w.contents
        for nm, ln in (("u", l), ("v", l), ("u_t", l), ("g", l - 1), ("h", l - 1), ("b", m), ("c", n),
                       ("u_avgcon", l), ("v_avgcon", l)):
            work[nm] = np.ctypeslib.as_array(getattr(W, nm), shape=(ln,)).copy()
        for nm in ("mu", "beta", "sigma", "gamma", "g_th", "sc_b", "sc_c"):
            work[nm] = getattr(W, nm)
        L.abip_finish(w)
    after = {k: getattr(P.stgs, k) for k in ("avg_criterion", "dynamic_sigma", "max_admm_iters")}
    return Result(x, y, s, _info_dict(info), work, tr, after)


def state_after(kind: str, A, b, c, T: int, linsys: str = "indirect", **settings) -> Result:
    """State (u, v, u_t) after exactly T inner ADMM iterations of the FIRST outer iteration
    (max_admm_iters=T ends the inner loop after T passes unless the inner criterion fires first)."""
    settings = dict(settings)
    settings["max_admm_iters"] = T
    return solve(kind, A, b, c, linsys=linsys, **settings)
