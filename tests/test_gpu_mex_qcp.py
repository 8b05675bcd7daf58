"""GPU (-m gpu): the conic Matlab gateways of mex/abip_hip_qcp_mex.c EXECUTED against libabip_hip.so through the mock mex runtime:
[sol, info] = abip_qcp(data, cones, settings) on the reference's literal toy problem (recorded reference output),
[sol, info] = abip_ml(data, settings) for LASSO and both SVM formulations against the Python mirror of the same surface."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from _lasso_cases import gen as lasso_gen
from _svm_cases import gen as svm_gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INFO_FIELDS = ["ipm_iter", "admm_iter", "status", "pobj", "dobj", "res_pri", "res_dual", "gap", "status_val", "setup_time", "solve_time", "runtime",
               "lin_sys_time_per_iter", "avg_cg_iters"]   # abip_qcp_mex.c:121-124


@pytest.fixture(scope="module")
def product():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
    from abip_amd import _lib, qcp
    _lib.load()
    return qcp


def _gateway(name):
    G = C.CDLL(os.path.join(ROOT, "tests", "mock_mex", name))
    P = C.c_void_p
    G.mock_dense.restype = P; G.mock_dense.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]
    G.mock_sparse.restype = P; G.mock_sparse.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
    G.mock_struct.restype = P
    G.mock_struct_add.argtypes = [P, C.c_char_p, P]
    G.mock_calln.restype = C.c_int; G.mock_calln.argtypes = [C.c_int, C.POINTER(P), C.c_int, C.POINTER(P)]
    G.mock_last_error.restype = C.c_char_p
    G.mock_numel.restype = C.c_size_t; G.mock_numel.argtypes = [P]
    G.mock_data.restype = C.POINTER(C.c_double); G.mock_data.argtypes = [P]
    G.mock_string.restype = C.c_char_p; G.mock_string.argtypes = [P]
    G.mock_field.restype = P; G.mock_field.argtypes = [P, C.c_char_p]
    G.mock_nfields.restype = C.c_int; G.mock_nfields.argtypes = [P]
    G.mock_field_name.restype = C.c_char_p; G.mock_field_name.argtypes = [P, C.c_int]
    return G


def _dense(G, v, row=False):
    v = np.ascontiguousarray(np.atleast_1d(np.asarray(v, dtype=np.float64)))
    return G.mock_dense(1 if row else v.size, v.size if row else 1, v.ctypes.data_as(C.POINTER(C.c_double)))


def _sparse(G, M):
    M = sp.csc_matrix(M); M.sort_indices()
    jc = M.indptr.astype(np.uint64); ir = M.indices.astype(np.uint64); pr = M.data.astype(np.float64)
    return G.mock_sparse(M.shape[0], M.shape[1], jc.ctypes.data_as(C.POINTER(C.c_size_t)), ir.ctypes.data_as(C.POINTER(C.c_size_t)), pr.ctypes.data_as(C.POINTER(C.c_double)))


def _struct(G, fields):
    s = G.mock_struct()
    for k, v in fields.items():
        G.mock_struct_add(s, k.encode(), v)
    return s


def _read_struct(G, s):
    out = {}
    for f in range(G.mock_nfields(s)):
        nm = G.mock_field_name(s, f).decode()
        fld = G.mock_field(s, nm.encode())
        if not fld:
            out[nm] = None
        elif G.mock_string(fld):
            out[nm] = G.mock_string(fld).decode()
        else:
            out[nm] = np.ctypeslib.as_array(G.mock_data(fld), shape=(G.mock_numel(fld),)).copy() if G.mock_numel(fld) else np.zeros(0)
    return out


def _run(G, args, nlhs=2):
    out = (C.c_void_p * 2)()
    arr = (C.c_void_p * len(args))(*args)
    rc = G.mock_calln(nlhs, out, len(args), arr)
    if rc:
        return rc, G.mock_last_error().decode(), None
    sol = _read_struct(G, out[0])
    info = _read_struct(G, out[1])
    info = {k: (v if isinstance(v, str) else float(v[0])) for k, v in info.items()}
    return 0, sol, info


def test_abip_qcp_gateway_on_the_toy_problem(product):
    G = _gateway("libmexgw_hip_qcp.so")
    A = np.array([[1, 2, 3, 4, 5, 6, 7, 8], [0, 1, 2, 1, 2, 3, 1, 2]], dtype=float)
    data = _struct(G, dict(A=_sparse(G, A), Q=_sparse(G, sp.identity(8)), b=_dense(G, [4.0, 3.0]), c=_dense(G, [1, 0, 2, 1, 4, 2, 3, 0])))
    cones = _struct(G, dict(q=_dense(G, [3], row=True), rq=_dense(G, [3]), f=_dense(G, [1]), l=_dense(G, [1])))
    stg = _struct(G, dict(eps=_dense(G, [1e-6]), linsys_solver=_dense(G, [1]), verbose=_dense(G, [0])))
    rc, sol, info = _run(G, [data, cones, stg])
    assert rc == 0
    assert list(info.keys()) == INFO_FIELDS and list(sol.keys()) == ["x", "y", "s"]
    assert info["status"] == "Solved" and info["ipm_iter"] == 10 and info["admm_iter"] == 91 and info["status_val"] == 1   # the reference's recorded run
    assert abs(info["pobj"] - (-0.984063813)) < 5e-9 and abs(info["dobj"] - (-0.984063938)) < 5e-9
    want = np.array([0.046341, 0.044938, 0.011319, 0.342543, 0.061490, 0.205246, -2.161307, 2.006235])
    assert sol["x"].size == 8 and sol["y"].size == 2 and sol["s"].size == 8 and np.max(np.abs(sol["x"] - want)) < 6e-7
    assert abs(info["runtime"] - (info["setup_time"] + info["solve_time"])) < 1e-9
    # the gateway's own argument checks (abip_qcp_mex.c:136-214)
    rc, msg, _ = _run(G, [_struct(G, dict(A=_sparse(G, A), b=_dense(G, [4.0, 3.0]))), cones, stg])
    assert rc == 1 and "must contain a `c` entry" in msg
    rc, msg, _ = _run(G, [_struct(G, dict(A=_dense(G, [1.0]), b=_dense(G, [4.0]), c=_dense(G, [1.0]))), cones, stg])
    assert rc == 1 and "sparse format" in msg
    rc, msg, _ = _run(G, [data, stg])
    assert rc == 1 and "data struct, cone struct, settings struct" in msg


def test_abip_ml_gateway_lasso_and_svm(product):
    G = _gateway("libmexgw_hip_ml.so")
    X, y, lam = lasso_gen("wide_sparse")
    mk = lambda X_, y_, l_: _struct(G, {"X": _sparse(G, X_), "y": _dense(G, y_), "lambda": _dense(G, [l_])})
    st = lambda pt: _struct(G, dict(prob_type=_dense(G, [pt]), eps=_dense(G, [1e-4]), linsys_solver=_dense(G, [1]), verbose=_dense(G, [0])))
    rc, sol, info = _run(G, [mk(X, y, lam), st(0)])
    assert rc == 0 and list(sol.keys()) == ["x"] and list(info.keys()) == INFO_FIELDS and info["status"] == "Solved"
    want, wi = product.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, eps=1e-4, linsys_solver=1, verbose=0))
    assert info["admm_iter"] == wi["admm_iter"] and np.array_equal(sol["x"], want["x"])            # same library, same inputs: bit-identical
    Xs, ys = svm_gen("tall")
    C_ = 1.0 / (Xs.shape[0] * 1e-2)
    for pt, lm in ((1, C_), (3, 1e-2)):
        rc, sol, info = _run(G, [mk(Xs, ys, lm), st(pt)])
        assert rc == 0 and list(sol.keys()) == ["w", "b", "xi", "x"] and info["status"] == "Solved"
        want, wi = product.abip_ml(dict(X=Xs, y=ys, **{"lambda": lm}), dict(prob_type=pt, eps=1e-4, linsys_solver=1, verbose=0))
        assert info["admm_iter"] == wi["admm_iter"] and np.array_equal(sol["w"], want["w"]) and sol["b"][0] == want["b"] and np.array_equal(sol["xi"], want["xi"])
        assert np.array_equal(sol["x"], sol["w"]) and sol["w"].size == Xs.shape[1] and sol["xi"].size == Xs.shape[0]
    # argument checks (abip_ml_mex.c:117-144, 266-276)
    rc, msg, _ = _run(G, [mk(X, y, lam), _struct(G, dict(eps=_dense(G, [1e-3])))])
    assert rc == 1 and "problem type" in msg
    rc, msg, _ = _run(G, [mk(X, y, lam), st(2)])
    assert rc == 1 and "Invalid problem type" in msg
    rc, msg, _ = _run(G, [_struct(G, {"X": _sparse(G, X), "y": _dense(G, y)}), st(0)])
    assert rc == 1 and "`lambda`" in msg
