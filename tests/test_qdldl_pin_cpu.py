"""CPU: the one piece of the conic path that can be held against REFERENCE CODE in this image.  The reference's conic sources need MKL headers
(src/abip-qcp/include/cones.h:11-12, linsys.h:14-18) and cannot be built here; its vendored QDLDL (src/external/qdldl/src/qdldl.c: QDLDL_factor /
QDLDL_solve, the conic KKT solve of source/linsys.c:310-316) can.  tests/golden/qdldl_*.npz hold QDLDL's solutions and pivots (generated in the container
by tests/golden/make_golden_qdldl.py from oracle/_ref/libqdldl_ref.so) on KKT matrices of the conic path's shape; here
  (i)  the oracle's own LDL' (oracle/orc_ldl.h, used by oracle/abip_qcp_oracle.c for every conic solve) and
  (ii) the product's host factorisation (host_setup.cpp: ordering + head + Schur complement + dense tail, every tail choice)
solve the same systems to 1e-11, and where the live QDLDL build is present it reproduces its own fixtures bit for bit.
(The conic path as a whole -- a13 .. a15, f4 -- stays "partial": only its linear solve is pinned.)"""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(os.path.basename(f)[6:-4] for f in glob.glob(os.path.join(ROOT, "tests", "golden", "qdldl_*.npz")))
pi, pf = C.POINTER(C.c_int), C.POINTER(C.c_double)


def load(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"qdldl_{name}.npz"))
    return int(z["n"]), z["Up"].astype(np.int32), z["Ui"].astype(np.int32), z["Ux"].astype(np.float64), z["B"], z["X"], z["D"]


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_fixtures_exist():
    assert set(CASES) >= {"toy", "lasso_small", "rsoc_mix", "lp_kkt"}


@pytest.mark.parametrize("name", CASES)
def test_fixture_solves_its_system_and_has_the_quasi_definite_inertia(name):
    n, Up, Ui, Ux, B, X, D = load(name)
    U = sp.csc_matrix((Ux, Ui, Up), shape=(n, n))
    K = U + sp.triu(U, 1).T
    for k in range(B.shape[0]):
        assert rel(K @ X[k], B[k]) < 1e-13
    assert (D != 0).all() and 0 < (D < 0).sum() < n


@pytest.mark.parametrize("name", CASES)
def test_oracle_ldl_against_qdldl(name):
    from oracle import pyoracle_qcp as pq
    L = pq.lib()
    L.orc_ldl_solve_upper.argtypes = [C.c_int, pi, pi, pf, pf]
    L.orc_ldl_solve_upper.restype = C.c_int
    n, Up, Ui, Ux, B, X, D = load(name)
    for k in range(B.shape[0]):
        b = B[k].copy()
        assert L.orc_ldl_solve_upper(n, Up.ctypes.data_as(pi), Ui.ctypes.data_as(pi), Ux.ctypes.data_as(pf), b.ctypes.data_as(pf)) == 0
        assert rel(b, X[k]) < 1e-11


@pytest.mark.parametrize("tail", [-1, 0, 64])
@pytest.mark.parametrize("name", CASES)
def test_product_host_factor_against_qdldl(name, tail):
    from abip_amd import _lib
    L = _lib.load()
    L.abip_hip_ldl_solve.argtypes = [C.c_int, pi, pi, pf, C.c_int, C.c_int, pf, pf]
    L.abip_hip_ldl_solve.restype = C.c_int
    n, Up, Ui, Ux, B, X, D = load(name)
    if tail > n:
        pytest.skip("tail larger than the system")
    st = np.zeros(4)
    for k in range(B.shape[0]):
        b = B[k].copy()
        assert L.abip_hip_ldl_solve(n, Up.ctypes.data_as(pi), Ui.ctypes.data_as(pi), Ux.ctypes.data_as(pf), tail, 0, b.ctypes.data_as(pf), st.ctypes.data_as(pf)) == 0
        assert rel(b, X[k]) < 1e-11
    if tail == 0:
        assert st[0] == 0


@pytest.mark.parametrize("name", CASES)
def test_live_qdldl_reproduces_its_fixture(name):
    so = os.path.join(ROOT, "oracle", "_ref", "libqdldl_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libqdldl_ref.so is built where the reference tree exists")
    Q = C.CDLL(so)
    Q.qdldl_ref_solve.argtypes = [C.c_int, pi, pi, pf, pf, pf]
    Q.qdldl_ref_solve.restype = C.c_int
    n, Up, Ui, Ux, B, X, D = load(name)
    d = np.zeros(n)
    for k in range(B.shape[0]):
        b = B[k].copy()
        assert Q.qdldl_ref_solve(n, Up.ctypes.data_as(pi), Ui.ctypes.data_as(pi), Ux.ctypes.data_as(pf), b.ctypes.data_as(pf), d.ctypes.data_as(pf)) == 0
        assert np.array_equal(b, X[k]) and np.array_equal(d, D)
