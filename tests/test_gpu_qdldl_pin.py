"""GPU (-m gpu): the device LDL' of the direct back-ends (host head + device dense tail, dev_ldl.h) against the reference's QDLDL on the fixtures of
tests/golden/qdldl_*.npz (see tests/test_qdldl_pin_cpu.py): 1e-11, with and without a dense tail."""
import ctypes as C

import numpy as np
import pytest

from test_qdldl_pin_cpu import CASES, load, pf, pi, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tail", [-1, 0, 64])
@pytest.mark.parametrize("name", CASES)
def test_device_ldl_against_qdldl(name, tail):
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
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
        assert L.abip_hip_ldl_solve(n, Up.ctypes.data_as(pi), Ui.ctypes.data_as(pi), Ux.ctypes.data_as(pf), tail, 1, b.ctypes.data_as(pf), st.ctypes.data_as(pf)) == 0
        assert rel(b, X[k]) < 1e-11, (name, tail, st)
    if tail == 64:
        assert st[0] == 64
