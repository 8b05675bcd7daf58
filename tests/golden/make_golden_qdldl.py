"""Container only: the reference's QDLDL (oracle/_ref/libqdldl_ref.so = src/external/qdldl/src/qdldl.c behind oracle/qdldl_ref_driver.c) on KKT matrices of the
conic path's shape  K = [[Q + rho_x I, A'], [A, -rho_y I]]  (qcp_config.c:699-748 up to the sign / block order the factorisation does not care about) for the
reference's toy problem (test/test_abip_install.m:32-43), a small LASSO-as-SOCP and a mixed SOC / RSOC / free / zero / orthant problem, plus one LP KKT matrix.
Writes tests/golden/qdldl_<name>.npz: the upper triangle (CSC), three right-hand sides, QDLDL's solutions and pivots.  Data only."""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def kkt_upper(A, Q, rho_x, rho_y):
    m, n = A.shape
    Q = sp.csc_matrix((n, n)) if Q is None else sp.csc_matrix(Q)
    K = sp.bmat([[Q + rho_x * sp.identity(n), A.T], [A, -rho_y * sp.identity(m)]], format="csc")
    U = sp.triu(K, format="csc")
    U.sort_indices()
    return U


def qdldl(U, B):
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libqdldl_ref.so"))
    pi, pf = C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.qdldl_ref_solve.argtypes = [C.c_int, pi, pi, pf, pf, pf]
    L.qdldl_ref_solve.restype = C.c_int
    n = U.shape[0]
    Ap, Ai, Ax = U.indptr.astype(np.int32), U.indices.astype(np.int32), U.data.astype(np.float64)
    X = np.zeros_like(B)
    D = np.zeros(n)
    for k in range(B.shape[0]):
        x = B[k].copy()
        rc = L.qdldl_ref_solve(n, Ap.ctypes.data_as(pi), Ai.ctypes.data_as(pi), Ax.ctypes.data_as(pf), x.ctypes.data_as(pf), D.ctypes.data_as(pf))
        assert rc == 0
        X[k] = x
    return X, D


def cases():
    from qcp_cases import lasso_socp, mixed   # the problems the conic tests use
    rng = np.random.default_rng(2026)
    Atoy = sp.csc_matrix(np.array([[1, 2, 3, 4, 5, 6, 7, 8], [0, 1, 2, 1, 2, 3, 1, 2]], dtype=float))    # test/test_abip_install.m:32-43
    yield "toy", kkt_upper(Atoy, sp.identity(8, format="csc"), 1.0, 1e-6)
    d, K = lasso_socp(60, 150, 3, density=0.3)
    yield "lasso_small", kkt_upper(sp.csc_matrix(d["A"]), d.get("Q"), 1.0, 1e-6)
    d, K = mixed(5)
    yield "rsoc_mix", kkt_upper(sp.csc_matrix(d["A"]), d.get("Q"), 1.0, 1e-6)
    A = sp.random(40, 90, density=0.15, random_state=rng, data_rvs=rng.standard_normal, format="csc") + sp.hstack([sp.identity(40), sp.csc_matrix((40, 50))])
    yield "lp_kkt", kkt_upper(sp.csc_matrix(A), None, 1.0, 1e-3)       # [[I, A'],[A, -rho I]]: the LP path's matrix up to block order and sign


if __name__ == "__main__":
    rng = np.random.default_rng(7)
    for name, U in cases():
        n = U.shape[0]
        B = rng.standard_normal((3, n))
        X, D = qdldl(U, B)
        Kfull = U + sp.triu(U, 1).T
        res = max(np.linalg.norm(Kfull @ X[k] - B[k]) / np.linalg.norm(B[k]) for k in range(3))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"qdldl_{name}.npz"), n=n, Up=U.indptr.astype(np.int32), Ui=U.indices.astype(np.int32), Ux=U.data,
                            B=B, X=X, D=D)
        print(f"{name}: n {n}, nnz(upper) {U.nnz}, QDLDL residual {res:.2e}, negative pivots {(D < 0).sum()}")
