"""Thin object wrapper over the C ABI: one `Solver` == one ABIPWork (abip_init .. abip_finish)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import ABIPData, ABIPInfo, ABIPMatrix, ABIPSettings, ABIPSolution, AbipHipProfile, K_CLASSES, PF, PI, c_int

__all__ = ["Solver", "default_settings", "LINSYS_DIRECT", "LINSYS_INDIRECT"]

LINSYS_DIRECT, LINSYS_INDIRECT = 0, 1

# src/abip-lp/src/util.c:288-329; max_time / pfeasopt are set by the mex only (abip_mex.c:320-341)
_EXTRA_DEFAULTS = dict(max_time=3600.0, pfeasopt=0)


def default_settings(**over) -> ABIPSettings:
    L = _lib.load()
    s = ABIPSettings()
    d = ABIPData()
    d.stgs = C.pointer(s)
    L.abip_set_default_settings(C.byref(d))
    for k, v in {**_EXTRA_DEFAULTS, **over}.items():
        if not hasattr(s, k):
            raise KeyError(f"unknown ABIP setting {k!r}")
        setattr(s, k, v)
    return s


def _f(a):
    return a.ctypes.data_as(PF)


def _i(a):
    return a.ctypes.data_as(PI)


def info_dict(info: ABIPInfo) -> dict:
    d = {k: getattr(info, k) for k, _ in ABIPInfo._fields_ if k != "status"}
    d["status"] = info.status.decode()
    return d


class Solver:
    """Owns host copies of (A, b, c) and the device work (abip_init scales a private copy of A unless abip_hip_set_copy_a_matrix(0))."""

    def __init__(self, A, b, c, linsys: int | str = LINSYS_DIRECT, **settings):
        self.w = None
        self.L = _lib.load()
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.m, self.n = A.shape
        self.Ax = np.array(A.data, dtype=np.float64, copy=True)
        self.Ai = np.array(A.indices, dtype=np.int64, copy=True)
        self.Ap = np.array(A.indptr, dtype=np.int64, copy=True)
        self.b = np.array(b, dtype=np.float64, copy=True)
        self.c = np.array(c, dtype=np.float64, copy=True)
        self.stgs = default_settings(**settings)
        self.mat = ABIPMatrix(_f(self.Ax), _i(self.Ai), _i(self.Ap), self.m, self.n)
        self.data = ABIPData(self.m, self.n, C.pointer(self.mat), _f(self.b), _f(self.c),
                             float(self.Ax.size) / (float(self.m) * float(self.n)), C.pointer(self.stgs))
        self.info = ABIPInfo()
        if isinstance(linsys, str):
            linsys = {"direct": LINSYS_DIRECT, "indirect": LINSYS_INDIRECT, "pcg": LINSYS_INDIRECT}[linsys]
        self.linsys = int(linsys)
        self.L.abip_hip_set_linsys(self.linsys)
        self.w = self.L.abip_init(C.byref(self.data), C.byref(self.info))
        if not self.w:
            raise RuntimeError("abip_init failed (invalid data, factorisation failure or no usable HIP device)")
        self.x = np.full(self.n, np.nan)
        self.y = np.full(self.m, np.nan)
        self.s = np.full(self.n, np.nan)
        self.sol = ABIPSolution(_f(self.x), _f(self.y), _f(self.s))
        self.finished = False

    # -- the reference's call ------------------------------------------------------------------
    def solve(self) -> dict:
        self.L.abip_solve(self.w, C.byref(self.data), C.byref(self.sol), C.byref(self.info))
        self.finished = True
        return info_dict(self.info)

    # -- stepping -------------------------------------------------------------------------------
    def begin(self, warm: Optional[tuple] = None) -> None:
        if warm is not None:
            self.x[:], self.y[:], self.s[:] = warm
        rc = self.L.abip_hip_solve_begin(self.w, C.byref(self.data), C.byref(self.sol), C.byref(self.info))
        if rc != 0:
            raise RuntimeError("abip_hip_solve_begin failed")
        self.finished = False

    def step(self, nsteps: int) -> tuple[bool, int]:
        done = c_int(0)
        fin = self.L.abip_hip_step(self.w, int(nsteps), C.byref(done), C.byref(self.info))
        self.finished = bool(fin)
        return self.finished, int(done.value)

    def end(self) -> dict:
        self.L.abip_hip_solve_end(self.w, C.byref(self.sol), C.byref(self.info))
        return info_dict(self.info)

    def sync(self) -> None:
        self.L.abip_hip_sync(self.w)

    # -- unit-level kernels ---------------------------------------------------------------------
    def accum_by_A(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.array(y, dtype=np.float64, copy=True)
        if self.L.abip_hip_accum_by_A(self.w, _f(x), _f(y)) != 0:
            raise RuntimeError("abip_hip_accum_by_A failed")
        return y

    def accum_by_Atrans(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.array(y, dtype=np.float64, copy=True)
        if self.L.abip_hip_accum_by_Atrans(self.w, _f(x), _f(y)) != 0:
            raise RuntimeError("abip_hip_accum_by_Atrans failed")
        return y

    def kkt_solve(self, rhs, warm=None, it: int = -1):
        rhs = np.array(rhs, dtype=np.float64, copy=True)
        wp = None
        if warm is not None:
            warm = np.ascontiguousarray(warm, dtype=np.float64)
            wp = _f(warm)
        its = self.L.abip_hip_kkt_solve(self.w, _f(rhs), wp, int(it))
        if its < 0:
            raise RuntimeError("abip_hip_kkt_solve failed")
        return rhs, int(its)

    def vector(self, name: str) -> np.ndarray:
        cap = max(self.m + self.n + 1, self.Ax.size)
        out = np.zeros(cap)
        ln = self.L.abip_hip_get_vector(self.w, name.encode(), _f(out), out.size)
        if ln < 0:
            raise KeyError(name)
        return out[:ln].copy()

    def rows(self) -> tuple[int, int]:
        """This rank's row range [row0, row1) of the sharded PCG path ((0, m) on a single GPU)."""
        r0, r1 = c_int(0), c_int(0)
        self.L.abip_hip_dist_rows(self.w, C.byref(r0), C.byref(r1))
        return int(r0.value), int(r1.value)

    def scalar(self, name: str) -> float:
        return float(self.L.abip_hip_get_scalar(self.w, name.encode()))

    # -- measurement ----------------------------------------------------------------------------
    def profile_enable(self, classes=K_CLASSES) -> None:
        mask = 0
        for cname in classes:
            mask |= 1 << (len(K_CLASSES) if cname == "allreduce" else K_CLASSES.index(cname))   # "allreduce": the collectives of a sharded solve
        self.L.abip_hip_profile_enable(self.w, mask)

    def profile_enable_stamps(self, classes=("spmv_At", "spmv_A")) -> None:
        """Device-side begin/end ticks of the PCG SpMV launches (no event records; cheap enough for a timed region)."""
        mask = 0
        for cname in classes:
            mask |= 1 << K_CLASSES.index(cname)
        if self.L.abip_hip_profile_enable_stamps(self.w, mask) != 0:
            raise RuntimeError("abip_hip_profile_enable_stamps failed")

    def profile_read(self, reset: bool = True) -> dict:
        p = AbipHipProfile()
        self.L.abip_hip_profile_read(self.w, C.byref(p), 1 if reset else 0)
        return dict(ms={k: p.ms[i] for i, k in enumerate(K_CLASSES)},
                    launches={k: p.launches[i] for i, k in enumerate(K_CLASSES)},
                    noop_ms=p.noop_ms, noop_launches=p.noop_launches,
                    admm_iters=p.admm_iters, cg_iters=p.cg_iters, kkt_solves=p.kkt_solves,
                    stamp_ms={k: p.stamp_ms[i] for i, k in enumerate(K_CLASSES)},
                    stamp_launches={k: p.stamp_launches[i] for i, k in enumerate(K_CLASSES)},
                    stamp_noop_launches=p.stamp_noop_launches,
                    allreduce_ms=p.allreduce_ms, allreduce_calls=p.allreduce_calls, allreduce_bytes=p.allreduce_bytes,
                    cg_iters_skipped=p.cg_iters_skipped)

    def close(self) -> None:
        if getattr(self, "w", None):
            self.L.abip_finish(self.w)
            self.w = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
