"""CPU: the host half of the persistent launch for cache-resident LPs (abip_amd/csrc/solver.hip xcd_plan, dev_xcd.h) without a GPU, through the
pure-host entry point abip_hip_xcd_plan: which problems are admitted, on how many workgroups / XCDs, that the slices of A and A' partition
the rows, fit the kernel variant's registers (NZ non-zeros and RM / RN rows per thread, 768 threads) and the 160 KB of LDS."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from abip_amd import _lib, problems

XTB = 768     # threads per workgroup (dev_xcd.h)


def plan(A, linsys, G=None, monkeypatch=None):
    L = _lib.load()
    A = sp.csc_matrix(A); A.sort_indices()
    Ai = np.ascontiguousarray(A.indices, dtype=np.int64); Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
    out = np.zeros(8); mb = np.zeros(257, dtype=np.int32); nb = np.zeros(257, dtype=np.int32)
    pi32 = C.POINTER(C.c_int)
    L.abip_hip_xcd_plan.restype = C.c_int
    L.abip_hip_xcd_plan.argtypes = [C.c_long, C.c_long, _lib.PI, _lib.PI, C.c_int, _lib.PF, pi32, pi32]
    rc = L.abip_hip_xcd_plan(A.shape[0], A.shape[1], Ap.ctypes.data_as(_lib.PI), Ai.ctypes.data_as(_lib.PI), 1 if linsys == "indirect" else 0,
                             out.ctypes.data_as(_lib.PF), mb.ctypes.data_as(pi32), nb.ctypes.data_as(pi32))
    assert rc == 0
    g = int(out[1])
    return dict(ok=bool(out[0]), G=g, xcds=int(out[2]), NZ=int(out[3]), RM=int(out[4]), RN=int(out[5]), lds=int(out[6]), minv_rows=int(out[7]),
                mb=mb[:g + 1].copy(), nb=nb[:g + 1].copy())


def check_slices(A, p):
    A = sp.csr_matrix(A); At = sp.csr_matrix(A.T)
    for M, cuts, R in ((A, p["mb"], p["RM"]), (At, p["nb"], p["RN"])):
        assert cuts[0] == 0 and cuts[-1] == M.shape[0] and np.all(np.diff(cuts) >= 0)           # a partition of the rows, in order
        nnz = np.diff(M.indptr[cuts])
        assert nnz.max() <= p["NZ"] * XTB and np.diff(cuts).max() <= R * XTB                     # what a thread holds in registers
        assert np.diff(M.indptr).max() <= 512                                                    # a row is added up by one thread
    assert 84 * 1024 <= p["lds"] <= 160 * 1024                                                   # one workgroup per CU, and it fits


def test_linsys_constants():
    import re, os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "abip.h")).read()
    assert re.search(r"#define ABIP_HIP_LINSYS_DIRECT\s+0", hdr) and re.search(r"#define ABIP_HIP_LINSYS_INDIRECT\s+1", hdr)   # what plan() passes


@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_netlib_class_surrogates_run_on_one_xcd(linsys, monkeypatch):
    monkeypatch.delenv("ABIP_HIP_XCD_G", raising=False)
    A = problems.lp_staircase()[0]                       # the C2 surrogate: 816 x ~2000, ~1e4 non-zeros
    p = plan(A, linsys)
    assert p["ok"] and p["G"] == 32 and p["xcds"] == 1 and p["NZ"] == 2
    check_slices(A, p)
    if linsys == "direct":
        assert 1 <= p["minv_rows"] <= int(np.diff(p["mb"]).max())


def test_pds_class_surrogate_spreads_over_all_xcds(monkeypatch):
    monkeypatch.delenv("ABIP_HIP_XCD_G", raising=False)
    A = problems.lp_multicommodity(nodes=1200, arcs=4400, commodities=10)[0]      # the C3 surrogate: 136 k non-zeros
    p = plan(A, "indirect")
    assert p["ok"] and p["G"] == 256 and p["xcds"] == 8 and p["NZ"] == 2           # (533 non-zeros per slice, 512 threads; round 4 stopped at four XCDs)
    check_slices(A, p)
    assert not plan(A, "direct")["ok"]                   # m = 16 390: no dense inverse of the Schur complement


def test_pcg_workgroup_count_follows_the_non_zero_count(monkeypatch):
    """One XCD up to 48 k non-zeros, four up to 120 k, eight beyond (scripts/xcd_g_sweep.py: two never win)."""
    monkeypatch.delenv("ABIP_HIP_XCD_G", raising=False)
    for kw, G in ((dict(nodes=240, arcs=1100, commodities=12), 32), (dict(nodes=400, arcs=2000, commodities=16), 128), (dict(nodes=900, arcs=4400, commodities=24), 256)):
        A = problems.lp_multicommodity(**kw)[0]
        p = plan(A, "indirect")
        assert p["ok"] and p["G"] == G, (kw, p["G"], p["NZ"])
        check_slices(A, p)


@pytest.mark.parametrize("G", [32, 64, 128, 256])
def test_forced_workgroup_counts(G, monkeypatch):
    monkeypatch.setenv("ABIP_HIP_XCD_G", str(G))
    A = problems.lp_multicommodity(nodes=200, arcs=700, commodities=5)[0]
    p = plan(A, "indirect")
    assert p["ok"] and p["G"] == G and p["xcds"] == G // 32
    check_slices(A, p)
    tiny = problems.lp_afiro_like()[0]                   # fewer rows than workgroups: empty slices are fine
    q = plan(tiny, "indirect")
    assert q["ok"] and q["G"] == G
    check_slices(tiny, q)


def test_direct_back_end_spreads_out_when_the_dense_inverse_dominates(monkeypatch):
    monkeypatch.delenv("ABIP_HIP_XCD_G", raising=False)
    small = problems.lp_staircase()[0]                                            # m = 816
    big = problems.lp_staircase(stages=20, rows_per=100, cols_per=230)[0]         # m = 2000
    p, q = plan(small, "direct"), plan(big, "direct")
    assert p["ok"] and p["G"] == 32 and q["ok"] and q["G"] == 256 and q["xcds"] == 8
    check_slices(big, q)
    assert q["minv_rows"] >= 1
    mid = problems.lp_staircase(stages=12, rows_per=100, cols_per=230)[0]         # m = 1200: the larger form of the dense product, four XCDs
    r = plan(mid, "direct")
    assert r["ok"] and r["G"] == 128 and r["xcds"] == 4
    check_slices(mid, r)


def test_what_does_not_fit_is_left_to_the_launch_path(monkeypatch):
    monkeypatch.delenv("ABIP_HIP_XCD_G", raising=False)
    A = problems.lp_random_sparse(m=20000, n=50000, per_col=16, seed=3)[0]       # 5e5 non-zeros: six per thread on 256 workgroups (round 4 left this one to the launch path)
    p = plan(A, "indirect")
    assert p["ok"] and p["G"] == 256 and p["NZ"] == 6
    check_slices(A, p)
    A = problems.lp_random_sparse(m=20000, n=50000, per_col=32, seed=3)[0]       # 9.8e5: no variant holds a slice -- the launch path's
    assert not plan(A, "indirect")["ok"]
    A = problems.lp_random_sparse(m=6500, n=13000, per_col=4, seed=4)[0]
    assert not plan(A, "direct")["ok"] and plan(A, "indirect")["ok"]             # direct: m > 6144
    rng = np.random.default_rng(0)
    D = sp.hstack([sp.csc_matrix(np.ones((1, 700))), sp.csc_matrix((1, 300))])  # one row of 700 entries: longer than a thread adds up
    A = sp.vstack([D, sp.random(60, 1000, density=0.01, random_state=rng, format="csc")]).tocsc()
    assert not plan(A, "indirect")["ok"]
