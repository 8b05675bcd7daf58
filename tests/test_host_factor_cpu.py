"""CPU: the host half of the direct back-end (abip_amd/csrc/host_setup.cpp) without a GPU -- minimum-degree ordering with lazy
updates, up-looking numeric factorisation of the head, Schur complement onto the dense tail, level-ordered storage -- through the
pure-host entry point abip_hip_host_factor_solve: K z = rhs must hold whatever part of the factor is declared "tail"."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from abip_amd import _lib, problems


def host_solve(A, rho, tail, rhs):
    L = _lib.load()
    A = sp.csc_matrix(A); A.sort_indices()
    Ax = np.ascontiguousarray(A.data, dtype=np.float64); Ai = np.ascontiguousarray(A.indices, dtype=np.int64); Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
    mat = _lib.ABIPMatrix(Ax.ctypes.data_as(_lib.PF), Ai.ctypes.data_as(_lib.PI), Ap.ctypes.data_as(_lib.PI), A.shape[0], A.shape[1])
    z = np.array(rhs, dtype=np.float64, copy=True)
    st = np.zeros(8)
    L.abip_hip_host_factor_solve.restype = C.c_int
    L.abip_hip_host_factor_solve.argtypes = [C.POINTER(_lib.ABIPMatrix), C.c_double, C.c_int, _lib.PF, _lib.PF]
    rc = L.abip_hip_host_factor_solve(C.byref(mat), float(rho), int(tail), z.ctypes.data_as(_lib.PF), st.ctypes.data_as(_lib.PF))
    assert rc == 0, rc
    return z, dict(N=int(st[0]), lnnz=int(st[1]), T=int(st[2]), levels=(int(st[3]), int(st[4])), head_nnz=int(st[5]))


def _wide_lasso(p, d):
    rng = np.random.default_rng(7)
    X = sp.random(p, d, density=min(1.0, 8.0 / p + 0.01), random_state=rng, data_rvs=rng.standard_normal, format="csc")
    r1 = sp.hstack([sp.csc_matrix(np.array([[1.0, -1.0]])), sp.csc_matrix((1, p + 2 * d))])
    r2 = sp.hstack([sp.csc_matrix((p, 2)), sp.identity(p), -X, X])
    return sp.vstack([r1, r2]).tocsc()


def kkt(A, rho):
    m, n = A.shape
    return sp.bmat([[rho * sp.identity(m), A], [A.T, -sp.identity(n)]], format="csc")


CASES = {
    "afiro": lambda: problems.lp_afiro_like()[0],
    "staircase": lambda: problems.lp_staircase()[0],
    "network": lambda: problems.lp_multicommodity()[0],
    "random": lambda: problems.lp_random_sparse(m=300, n=800, per_col=6, seed=5)[0],
    # few rows, many columns (a LASSO KKT with p << d): every row node ends up in hundreds of elements, the refreshed degrees use
    # the saturating sum bound -- the shape on which the lazy-update ordering once ran off the end of its degree lists
    "wide_lasso": lambda: _wide_lasso(20, 575),
    "wide_lasso2": lambda: _wide_lasso(69, 810),
    "dense_column": lambda: sp.hstack([problems.lp_random_sparse(m=200, n=500, per_col=3, seed=9)[0], sp.csc_matrix(np.ones((200, 1)))]).tocsc(),
}


@pytest.mark.parametrize("tail", [0, 64, 256, -1])
@pytest.mark.parametrize("name", sorted(CASES))
def test_factor_solves_the_kkt_system(name, tail):
    A = sp.csc_matrix(CASES[name]())
    m, n = A.shape
    rng = np.random.default_rng(3)
    rhs = rng.standard_normal(m + n)
    K = kkt(A, 1e-3)
    z, st = host_solve(A, 1e-3, tail, rhs)
    assert st["N"] == m + n
    assert np.linalg.norm(K @ z - rhs) <= 1e-10 * np.linalg.norm(rhs), (name, tail, st)
    if tail > 0:
        assert st["T"] == min(tail, (m + n - 1) // 64 * 64)
    if tail == 0:
        assert st["T"] == 0 and st["head_nnz"] == st["lnnz"]
    else:
        assert st["T"] % 64 == 0 and st["head_nnz"] <= st["lnnz"]


def test_tail_removes_the_sequential_levels_and_fill_is_sane():
    A = sp.csc_matrix(problems.lp_staircase()[0])
    m, n = A.shape
    rhs = np.ones(m + n)
    _, plain = host_solve(A, 1e-3, 0, rhs)
    _, auto = host_solve(A, 1e-3, -1, rhs)
    assert plain["levels"][0] > 100 and auto["levels"][0] <= 16 and auto["T"] >= 256        # hundreds of one-row levels -> a handful
    assert plain["lnnz"] == auto["lnnz"]                                                      # same ordering, same fill
    import scipy.sparse.linalg as spla
    lu = spla.splu(kkt(A, 1e-3).tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0)
    assert plain["lnnz"] <= 1.5 * lu.L.nnz                                                    # fill comparable to SuperLU's minimum-degree ordering


@pytest.mark.parametrize("variant", ["default", "origin", "qp", "scale5", "ruiz3"])
def test_host_scaling_is_bit_identical_to_the_oracle(oracle_built, variant):
    """ABIP(_normalize_A) (linsys/common.c:150-565): the product's host code against the oracle's restatement (itself bit-exact
    against the live reference, test_oracle_vs_ref.py) -- same D, E, scaled values and mean norms to the last bit."""
    po = oracle_built
    L = _lib.load()
    A0 = sp.csc_matrix(problems.lp_staircase()[0]); A0.sort_indices()
    m, n = A0.shape
    kw = dict(default={}, origin=dict(origin_rescale=1, pc_ruiz_rescale=0), qp=dict(qp_rescale=1, pc_ruiz_rescale=0), scale5=dict(scale=5.0), ruiz3=dict(ruiz_iter=3))[variant]
    # product
    Ax = np.array(A0.data, dtype=np.float64, copy=True); Ai = A0.indices.astype(np.int64); Ap = A0.indptr.astype(np.int64)
    D, E, means = np.zeros(m), np.zeros(n), np.zeros(2)
    mat = _lib.ABIPMatrix(Ax.ctypes.data_as(_lib.PF), Ai.ctypes.data_as(_lib.PI), Ap.ctypes.data_as(_lib.PI), m, n)
    st = _lib.ABIPSettings()
    dd = _lib.ABIPData(m, n, C.pointer(mat), None, None, 0.0, C.pointer(st))
    L.abip_set_default_settings(C.byref(dd))
    for k, v in kw.items():
        setattr(st, k, v)
    L.abip_hip_host_normalize_A.restype = C.c_int
    L.abip_hip_host_normalize_A.argtypes = [C.POINTER(_lib.ABIPMatrix), C.POINTER(_lib.ABIPSettings), _lib.PF, _lib.PF, _lib.PF]
    assert L.abip_hip_host_normalize_A(C.byref(mat), C.byref(st), D.ctypes.data_as(_lib.PF), E.ctypes.data_as(_lib.PF), means.ctypes.data_as(_lib.PF)) == 0
    # oracle
    P = po.Problem(A0, np.zeros(m), np.zeros(n), **kw)
    Do, Eo, mr, mc = np.zeros(m), np.zeros(n), np.zeros(1), np.zeros(1)
    po.lib("oracle").orc_normalize_A(C.byref(P.mat), C.byref(P.stgs), po._f(Do), po._f(Eo), po._f(mr), po._f(mc))
    assert np.array_equal(Ax, P.Ax) and np.array_equal(D, Do) and np.array_equal(E, Eo)
    assert means[0] == mr[0] and means[1] == mc[0]


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("name", ["staircase", "network", "wide_lasso2"])
def test_schur_hand_over_formats(name, mode, monkeypatch):
    """ABIP_HIP_DEV_SCHUR=1 / 2: the host stops before the Schur complement and hands over K22 as triplets plus L21 by columns (what k_schur_sub / k_schur_rows
    consume on the device); completed on the host here (host_setup.cpp: complete_schur_on_host), K z = rhs must hold as with the host's own accumulation."""
    A = sp.csc_matrix(CASES[name]())
    m, n = A.shape
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(m + n)
    K = kkt(A, 1e-3)
    monkeypatch.setenv("ABIP_HIP_DEV_SCHUR", "0")
    z0, st0 = host_solve(A, 1e-3, 128, rhs)
    monkeypatch.setenv("ABIP_HIP_DEV_SCHUR", mode)
    z1, st1 = host_solve(A, 1e-3, 128, rhs)
    assert st1["T"] == st0["T"] == 128 and st1["lnnz"] == st0["lnnz"]
    assert np.linalg.norm(K @ z1 - rhs) <= 1e-10 * np.linalg.norm(rhs)
    assert np.linalg.norm(z1 - z0) <= 1e-9 * np.linalg.norm(z0)
