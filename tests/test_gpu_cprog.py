"""GPU (-m gpu): the C boundary from a compiled C program -- abip_main, and two abip_solve calls on one
ABIPWork with new (b, c) (src/abip-lp/include/abip.h:116-124) -- and the 32-bit-index build of the library."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from _golden import info_of, load, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "abip_amd", "lib")


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
    exe = tmp_path_factory.mktemp("cprog") / "abip_c_prog"
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c", "abip_c_prog.c"), "-I", os.path.join(ROOT, "include"),
                    "-L", LIBDIR, "-labip_hip", f"-Wl,-rpath,{LIBDIR}", "-o", str(exe)], check=True)
    return str(exe)


def _write_problem(path, A, b, c, b2, c2):
    with open(path, "wb") as f:
        np.array([A.shape[0], A.shape[1], A.nnz], dtype=np.int64).tofile(f)
        A.indptr.astype(np.int64).tofile(f); A.indices.astype(np.int64).tofile(f); A.data.astype(np.float64).tofile(f)
        for v in (b, c, b2, c2):
            np.asarray(v, dtype=np.float64).tofile(f)


def _read_out(path, m, n, count):
    raw = np.fromfile(path, dtype=np.float64)
    per = 8 + 2 * n + m
    assert raw.size == per * count
    out = []
    for k in range(count):
        r = raw[k * per:(k + 1) * per]
        out.append((r[:8], r[8:8 + n], r[8 + n:8 + n + m], r[8 + n + m:]))
    return out


@pytest.mark.parametrize("linsys", ["direct", "indirect"])
def test_abip_main_from_c(prog, tmp_path, linsys):
    z, A, b, c = load("lp_afiro_like")
    pb, ob = str(tmp_path / "p.bin"), str(tmp_path / "o.bin")
    _write_problem(pb, A, b, c, b, c)
    env = dict(os.environ); env.pop("ABIP_HIP_LINSYS", None)
    p = subprocess.run([prog, pb, ob, "main", "1" if linsys == "indirect" else "0", "1e-6"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "2.0.0" in p.stdout
    (info, x, y, s), = _read_out(ob, A.shape[0], A.shape[1], 1)
    g = info_of(z, f"{linsys}_1e-06")
    assert info[0] == 1 and info[1] == g["ipm_iter"] and info[2] == g["admm_iter"]
    for got, k in ((x, "x"), (y, "y"), (s, "s")):
        assert rel(got, z[f"{linsys}_1e-06_{k}"]) < 1e-5, k
    assert abs(info[3] - g["pobj"]) <= 1e-5 * (1 + abs(g["pobj"]))


@pytest.mark.parametrize("linsys", ["direct", "indirect"])
def test_two_solves_on_one_work(prog, tmp_path, linsys):
    """The second abip_solve (new b, c; same A and ABIPWork) must give what a fresh abip_init + abip_solve gives on (A, b2, c2)."""
    import abip_amd
    z, A, b, c = load("lp_random_sparse_small")
    rng = np.random.default_rng(3)
    x0 = np.abs(rng.standard_normal(A.shape[1])) * (rng.random(A.shape[1]) < 0.4)
    b2 = A @ x0                                   # feasible by construction
    c2 = c * rng.uniform(0.5, 1.5, size=c.size)   # still positive: bounded
    pb, ob = str(tmp_path / "p.bin"), str(tmp_path / "o.bin")
    _write_problem(pb, A, b, c, b2, c2)
    p = subprocess.run([prog, pb, ob, "resolve", "1" if linsys == "indirect" else "0", "1e-6"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    first, second = _read_out(ob, A.shape[0], A.shape[1], 2)
    g = info_of(z, f"{linsys}_1e-06")
    assert first[0][0] == 1 and first[0][1] == g["ipm_iter"]
    for got, k in zip(first[1:], "xys"):
        assert rel(got, z[f"{linsys}_1e-06_{k}"]) < 1e-5, k
    with abip_amd.Solver(A, b2, c2, linsys=linsys, eps=1e-6, verbose=0) as S:
        fresh = S.solve()
        fx, fy, fs = S.x.copy(), S.y.copy(), S.s.copy()
    assert second[0][0] == fresh["status_val"] == 1
    assert second[0][1] == fresh["ipm_iter"] and second[0][2] == fresh["admm_iter"]      # same state machine from the same start: same counts
    for got, want, k in zip(second[1:], (fx, fy, fs), "xys"):
        assert rel(got, want) < 1e-9, k


def test_int32_library_solves(prog):
    """libabip_hip32.so: the same library built with -DABIP_INT32 (abip_int = int, the reference without DLONG, glbopts.h:86-94)."""
    path = os.path.join(LIBDIR, "libabip_hip32.so")
    assert os.path.exists(path)
    code = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from _golden import load, rel, info_of
from abip_amd import _lib
_lib._preload_hip_runtime()
L = C.CDLL(%r)
I, F = C.c_int, C.c_double
PF, PI = C.POINTER(F), C.POINTER(I)
class M(C.Structure): _fields_ = [("x", PF), ("i", PI), ("p", PI), ("m", I), ("n", I)]
names = [n for n, _ in _lib.ABIPSettings._fields_]
class S(C.Structure): _fields_ = [(n, I if t is _lib.c_int else F) for n, t in _lib.ABIPSettings._fields_]
class D(C.Structure): _fields_ = [("m", I), ("n", I), ("A", C.POINTER(M)), ("b", PF), ("c", PF), ("sp", F), ("stgs", C.POINTER(S))]
class Sol(C.Structure): _fields_ = [("x", PF), ("y", PF), ("s", PF)]
class Info(C.Structure): _fields_ = [("status", C.c_char * 32), ("status_val", I), ("ipm_iter", I), ("admm_iter", I)] + [(k, F) for k in ("pobj", "dobj", "res_pri", "res_dual", "rel_gap", "res_infeas", "res_unbdd", "setup_time", "solve_time")]
z, A, b, c = load("lp_afiro_like")
Ax = A.data.astype(np.float64).copy(); Ai = A.indices.astype(np.int32).copy(); Ap = A.indptr.astype(np.int32).copy()
b = b.astype(np.float64).copy(); c = c.astype(np.float64).copy()
m, n = A.shape
mat = M(Ax.ctypes.data_as(PF), Ai.ctypes.data_as(PI), Ap.ctypes.data_as(PI), m, n)
st = S(); d = D(m, n, C.pointer(mat), b.ctypes.data_as(PF), c.ctypes.data_as(PF), A.nnz / (m * n), C.pointer(st))
L.abip_set_default_settings(C.byref(d)); st.max_time = 3600.0; st.pfeasopt = 0; st.eps = 1e-6; st.verbose = 0
x = np.zeros(n); y = np.zeros(m); s = np.zeros(n)
sol = Sol(x.ctypes.data_as(PF), y.ctypes.data_as(PF), s.ctypes.data_as(PF)); info = Info()
L.abip_hip_set_linsys(0)
L.abip_main.restype = I
rc = L.abip_main(C.byref(d), C.byref(sol), C.byref(info))
g = info_of(z, "direct_1e-06")
assert rc == 1 and info.status_val == 1 and info.ipm_iter == g["ipm_iter"], (rc, info.status_val, info.ipm_iter)
for got, k in ((x, "x"), (y, "y"), (s, "s")):
    assert rel(got, z["direct_1e-06_" + k]) < 1e-5, k
print("INT32 OK", info.admm_iter)
''' % (ROOT, os.path.join(ROOT, "tests"), path)
    p = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "INT32 OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_in_place_scaling_switch(prog):
    """abip_hip_set_copy_a_matrix(0): the plain C build's behaviour -- A scaled in place by abip_init, un-scaled by abip_finish."""
    import abip_amd
    from abip_amd import _lib
    L = _lib.load()
    z, A, b, c = load("lp_afiro_like")
    try:
        L.abip_hip_set_copy_a_matrix(0)
        S = abip_amd.Solver(A, b, c, linsys="direct", eps=1e-4, verbose=0)
        during = S.Ax.copy()
        assert rel(during, S.vector("Ax")) < 1e-15 and rel(during, A.data) > 1e-3      # the caller's array holds the scaled matrix
        S.solve(); S.close()
        assert rel(S.Ax, A.data) < 1e-14                                                  # and gets it back (to rounding)
        L.abip_hip_set_copy_a_matrix(1)
        S = abip_amd.Solver(A, b, c, linsys="direct", eps=1e-4, verbose=0)
        assert np.array_equal(S.Ax, A.data)
        S.close()
    finally:
        L.abip_hip_set_copy_a_matrix(1)
