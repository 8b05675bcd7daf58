"""GPU (-m gpu): the Matlab gateways EXECUTED against libabip_hip.so through the mock mex runtime (tests/mock_mex):
  * mex/abip_hip_mex.c (this repository's gateway, abip_direct / abip_indirect builds),
  * the REFERENCE's own gateway src/abip-lp/mexfile/abip_mex.c, compiled in the build container from where it lies with the reference's
    headers and make_abip.m's flags into oracle/_ref/libmexgw_ref.so (travels to the GPU box; skipped where it was never built) --
    the drop-in claim of INTEGRATION.md section 1 end to end: [x, y, s, info] = abip_direct(data, params) returns the fixture."""
import ctypes as C
import os

import numpy as np
import pytest

from _golden import info_of, load, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INFO_FIELDS = ["status", "ipm_iter", "admm_iter", "mu", "pobj", "dobj", "resPri", "resDual", "relGap", "resInfeas", "resUnbdd", "setupTime", "solveTime"]


@pytest.fixture(scope="module")
def product():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
    from abip_amd import _lib
    return _lib.load()       # the product library first (and with it the one HIP runtime of the process)


def _gateway(path):
    G = C.CDLL(path)
    P = C.c_void_p
    G.mock_dense.restype = P; G.mock_dense.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]
    G.mock_sparse.restype = P; G.mock_sparse.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
    G.mock_struct.restype = P
    G.mock_struct_add.argtypes = [P, C.c_char_p, P]
    G.mock_call.restype = C.c_int; G.mock_call.argtypes = [C.c_int, C.POINTER(P), P, P]
    G.mock_last_error.restype = C.c_char_p
    G.mock_numel.restype = C.c_size_t; G.mock_numel.argtypes = [P]
    G.mock_data.restype = C.POINTER(C.c_double); G.mock_data.argtypes = [P]
    G.mock_string.restype = C.c_char_p; G.mock_string.argtypes = [P]
    G.mock_field.restype = P; G.mock_field.argtypes = [P, C.c_char_p]
    G.mock_nfields.restype = C.c_int; G.mock_nfields.argtypes = [P]
    G.mock_field_name.restype = C.c_char_p; G.mock_field_name.argtypes = [P, C.c_int]
    return G


def _call(G, A, b, c, params, nlhs=4, drop=()):
    def dense(v):
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(v, dtype=np.float64)))
        return G.mock_dense(v.size, 1, v.ctypes.data_as(C.POINTER(C.c_double)))
    jc = A.indptr.astype(np.uint64); ir = A.indices.astype(np.uint64); pr = A.data.astype(np.float64)
    data = G.mock_struct()
    A_mex = G.mock_sparse(A.shape[0], A.shape[1], jc.ctypes.data_as(C.POINTER(C.c_size_t)), ir.ctypes.data_as(C.POINTER(C.c_size_t)), pr.ctypes.data_as(C.POINTER(C.c_double)))
    if "A" not in drop:
        G.mock_struct_add(data, b"A", A_mex)
    G.mock_struct_add(data, b"b", dense(b)); G.mock_struct_add(data, b"c", dense(c))
    stg = G.mock_struct()
    for k, v in params.items():
        G.mock_struct_add(stg, k.encode(), dense(v))
    out = (C.c_void_p * 4)()
    rc = G.mock_call(nlhs, out, data, stg)
    if rc:
        return rc, G.mock_last_error().decode(), None
    vec = lambda a: np.ctypeslib.as_array(G.mock_data(a), shape=(G.mock_numel(a),)).copy()
    x, y, s = vec(out[0]), vec(out[1]), vec(out[2])
    info = {}
    for f in range(G.mock_nfields(out[3])):
        nm = G.mock_field_name(out[3], f).decode()
        fld = G.mock_field(out[3], nm.encode())
        info[nm] = None if not fld else (G.mock_string(fld).decode() if G.mock_string(fld) else float(G.mock_data(fld)[0]))
    # "Matlab's" matrix after the call: the gateway hands the library pointers INTO it (DLONG build, abip_mex.c:349-351)
    info["_A_after"] = np.ctypeslib.as_array(G.mock_data(A_mex), shape=(A.nnz,)).copy()
    return 0, (x, y, s), info


def _check(res, z, linsys, A):
    rc, (x, y, s), info = res
    assert rc == 0
    assert np.array_equal(info.pop("_A_after"), A.data)                   # COPYAMATRIX behaviour: the caller's matrix is never written
    g = info_of(z, f"{linsys}_1e-06")
    assert list(info.keys()) == INFO_FIELDS                               # abip_mex.c:101-102
    assert info["status"] == "Solved" and info["ipm_iter"] == g["ipm_iter"]
    assert info["admm_iter"] == g["admm_iter"]
    assert x.size == A.shape[1] and y.size == A.shape[0] and s.size == A.shape[1]
    for got, k in ((x, "x"), (y, "y"), (s, "s")):
        assert rel(got, z[f"{linsys}_1e-06_{k}"]) < 1e-5, k
    assert abs(info["pobj"] - g["pobj"]) <= 1e-5 * (1 + abs(g["pobj"]))
    assert abs(info["resPri"]) < 1e-6 and abs(info["relGap"]) < 1e-6


@pytest.mark.parametrize("linsys", ["direct", "indirect"])
def test_own_gateway_runs(product, linsys):
    G = _gateway(os.path.join(ROOT, "tests", "mock_mex", f"libmexgw_hip_{linsys}.so"))
    z, A, b, c = load("lp_afiro_like")
    _check(_call(G, A, b, c, dict(eps=1e-6, verbose=0, max_ipm_iters=500)), z, linsys, A)
    # the gateway's own argument checks (abip_mex.c:111-160)
    rc, msg, _ = _call(G, A, b, c, dict(eps=1e-6), drop=("A",))
    assert rc == 1 and "must contain a matrix 'A'" in msg
    rc, msg, _ = _call(G, A, b, c, dict(eps=1e-6), nlhs=5)
    assert rc == 1 and "up to 4 output" in msg


@pytest.mark.parametrize("linsys", ["direct", "indirect"])
def test_reference_gateway_runs_against_the_library(product, linsys):
    path = os.path.join(ROOT, "oracle", "_ref", "libmexgw_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libmexgw_ref.so was not built (no reference tree at build time)")
    G = _gateway(path)
    z, A, b, c = load("lp_afiro_like")
    product.abip_hip_set_linsys(1 if linsys == "indirect" else 0)       # the reference picks the back-end at link time; here: the one extra call of INTEGRATION.md
    _check(_call(G, A, b, c, dict(eps=1e-6, verbose=0)), z, linsys, A)
