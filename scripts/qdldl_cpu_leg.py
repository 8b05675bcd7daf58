#!/usr/bin/env python3
"""The kernel-level CPU leg of the conic direct back-end: the reference's own QDLDL_solve (src/external/qdldl/src/qdldl.c:236-281, compiled
from where it lies into oracle/_ref/libqdldl_ref.so) on the C5 KKT system, one thread.

    python scripts/qdldl_cpu_leg.py [--p 10000 --d 45000] [--nrhs 20] [--out profiles/r06_c5_cpu_qdldl_solve.json]

K = [[rho_x I + Q, A'], [A, -rho_y I]] of BASELINE configs[4] (LASSO-as-SOCP, n = 100 002, m = 10 001) in the elimination order [x | y] -- the Schur complement
onto the 10 001 rows is dense whatever the order, so L holds ~nnz(A) + m^2 / 2 = 5.5e7 non-zeros under any of them (the reference runs AMD first: linsys.c:272; the
product's own ordering ends on the same dense tail).  The one-off QDLDL_factor takes minutes; the figure wanted is the time of a SOLVE: what the device does in
~0.18 ms per ADMM iteration (profiles/r05_c5_direct_kernel_medians.txt).  Not a same-run baseline: bench.py quotes it under extra.cpu_solve_reference with the
date and the box, never under cpu_baseline."""
import argparse
import ctypes as C
import json
import os
import platform
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=int, default=10_000)
    ap.add_argument("--d", type=int, default=45_000)
    ap.add_argument("--nrhs", type=int, default=20)
    ap.add_argument("--rho-x", type=float, default=1.0)
    ap.add_argument("--rho-y", type=float, default=1e-3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from abip_amd import problems
    data, K = problems.qcp_lasso_socp(a.p, a.d)
    A = sp.csc_matrix(data["A"])
    m, n = A.shape
    Kkt = sp.bmat([[a.rho_x * sp.identity(n), A.T], [None, -a.rho_y * sp.identity(m)]], format="csc")   # upper triangle, diagonal present
    Kkt.sort_indices()
    N = n + m
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libqdldl_ref.so"))
    pi, pf = C.POINTER(C.c_int), C.POINTER(C.c_double)
    lib.qdldl_ref_time_solves.argtypes = [C.c_int, pi, pi, pf, C.c_int, pf, pf]
    lib.qdldl_ref_time_solves.restype = C.c_int
    Ap = np.ascontiguousarray(Kkt.indptr, dtype=np.int32); Ai = np.ascontiguousarray(Kkt.indices, dtype=np.int32); Ax = np.ascontiguousarray(Kkt.data, dtype=np.float64)
    rng = np.random.default_rng(3)
    B = rng.standard_normal((a.nrhs, N))
    B0 = B.copy()
    out = np.zeros(3)
    t0 = time.time()
    rc = lib.qdldl_ref_time_solves(N, Ap.ctypes.data_as(pi), Ai.ctypes.data_as(pi), Ax.ctypes.data_as(pf), a.nrhs, B.ctypes.data_as(pf), out.ctypes.data_as(pf))
    assert rc == 0, rc
    Kfull = Kkt + sp.triu(Kkt, 1).T
    res = max(np.linalg.norm(Kfull @ B[k] - B0[k]) / np.linalg.norm(B0[k]) for k in range(min(a.nrhs, 3)))
    rec = dict(what="QDLDL_solve of the reference (src/external/qdldl/src/qdldl.c:236-281, oracle/_ref/libqdldl_ref.so, gcc -O2) on the C5 KKT system, one thread",
               p=a.p, d=a.d, N=N, nnz_K_upper=int(Kkt.nnz), nnz_L=int(out[2]), factor_s=float(out[0]), solves=a.nrhs, solve_s_each=float(out[1] / a.nrhs),
               solves_per_s=float(a.nrhs / out[1]), relative_residual_max=float(res), threads=1, host_cores=os.cpu_count(),
               box=platform.node() + " / " + platform.processor() + " (" + next((ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?") + ")",
               date=time.strftime("%Y-%m-%d"), wall_s=time.time() - t0,
               bytes_streamed_per_solve=int(2 * 12 * out[2] + 3 * 8 * N))
    print(json.dumps(rec, indent=1))
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
