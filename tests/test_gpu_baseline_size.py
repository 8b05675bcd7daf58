"""GPU (-m gpu): the two configs the headline numbers are quoted on, at BASELINE size, against the REAL reference.

* C3 (pds-class multi-commodity LP, 16 390 x 48 400; bench.py's c3 workload): the reference's full solves at eps 1e-4 and 1e-6, both back-ends
  (tests/golden/lp_pds_like_full.npz, written by make_golden.py from oracle/_ref) -- the device must take the reference's EXACT number of outer and
  inner iterations on every path that serves this LP and land on its (x, y, s) within 10 eps.
* C4 (random sparse LP 200 000 x 500 000, the default bench workload): the reference stopped by max_admm_iters (tests/golden/lp_c4_prefix.npz: info,
  norms, sums and every 97th entry of x, y, s) -- same counts, same "Solved/Inaccurate", entries to 1e-8, as test_iteration_limits_and_inaccurate_status
  holds the small LP.  bench.py's CPU leg repeats the comparison live on the whole vectors (cpu_baseline.rel_err_xys).
The LPs are rebuilt by the seeded generators; the fixtures' checksum says whether they rebuilt the same ones."""
import numpy as np
import pytest
import scipy.sparse as sp

from _golden import GOLDEN, INFO_KEYS, lp_sha256, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    import abip_amd
    return abip_amd


@pytest.fixture(scope="module")
def c3():
    import os
    from abip_amd import problems
    z = np.load(os.path.join(GOLDEN, "lp_pds_like_full.npz"))
    A, b, c = problems.lp_multicommodity(nodes=1200, arcs=4400, commodities=10)
    A = sp.csc_matrix(A)
    assert lp_sha256(A, b, c) == str(z["lp_sha256"]), "the generator no longer rebuilds the LP the fixture was taken on"
    return z, A, b, c


@pytest.fixture(scope="module")
def c4():
    import os
    from abip_amd import problems
    z = np.load(os.path.join(GOLDEN, "lp_c4_prefix.npz"))
    A, b, c = problems.lp_random_sparse()
    assert lp_sha256(A, b, c) == str(z["lp_sha256"]), "the generator no longer rebuilds the LP the fixture was taken on"
    return z, A, b, c


@pytest.mark.parametrize("linsys,xcd", [("indirect", "1"), ("indirect", "0"), ("direct", "0")])
def test_c3_full_size_matches_the_reference(gpu, c3, linsys, xcd, monkeypatch):
    """abip.c:2131-2215 over a whole solve at BASELINE configs[2] size: persistent launch (PCG; four XCDs) and launch path (PCG; LDL' with a dense tail)."""
    z, A, b, c = c3
    monkeypatch.setenv("ABIP_HIP_XCD", xcd)
    for eps in (1e-4, 1e-6):
        tag = f"{linsys}_{eps:g}"
        g = dict(zip(INFO_KEYS, z[tag + "_info"]))
        with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=eps) as S:
            assert S.scalar("xcd") == float(xcd)
            info = S.solve()
            assert info["status_val"] == g["status_val"] == 1
            assert (info["ipm_iter"], info["admm_iter"]) == (g["ipm_iter"], g["admm_iter"]), (tag, xcd, info["ipm_iter"], info["admm_iter"], g["ipm_iter"], g["admm_iter"])
            for k in "xys":
                assert rel(getattr(S, k), z[f"{tag}_{k}"]) < 10 * eps, (tag, xcd, k)
            assert abs(info["pobj"] - g["pobj"]) <= 10 * eps * (1 + abs(g["pobj"])) and abs(info["dobj"] - g["dobj"]) <= 10 * eps * (1 + abs(g["dobj"]))
            for k in ("res_pri", "res_dual", "rel_gap"):
                assert info[k] < eps


@pytest.mark.parametrize("T", [25, 60])
def test_c4_prefix_matches_the_reference(gpu, c4, T):
    """The headline workload: the reference's run under max_admm_iters = T (its limit is looked at between inner loops and after each outer iteration, abip.c:2190-2245,
    so it stops a few iterations later: 33 at T = 25) against the device under the same cap -- equal counts and status, every 97th entry of (x, y, s) and their norms to 1e-8."""
    z, A, b, c = c4
    tag = f"indirect_T{T}"
    g = dict(zip(INFO_KEYS, z[tag + "_info"]))
    stride = int(z["stride"])
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-6, max_admm_iters=T) as S:
        info = S.solve()
        assert info["status"] == "Solved/Inaccurate" and info["status_val"] == g["status_val"]
        assert (info["ipm_iter"], info["admm_iter"]) == (g["ipm_iter"], g["admm_iter"]), (T, info["ipm_iter"], info["admm_iter"], g["ipm_iter"], g["admm_iter"])
        for k in "xys":
            v = getattr(S, k)
            assert rel(v[::stride], z[f"{tag}_{k}_sample"]) < 1e-8, (T, k)
            nrm, tot, big, where = z[f"{tag}_{k}_stats"]
            assert abs(np.linalg.norm(v) - nrm) <= 1e-8 * nrm and abs(v.sum() - tot) <= 1e-8 * max(abs(tot), nrm)
            assert abs(np.abs(v).max() - big) <= 1e-8 * big and int(np.argmax(np.abs(v))) == int(where)
        assert abs(info["pobj"] - g["pobj"]) <= 1e-8 * (1 + abs(g["pobj"])) and abs(info["dobj"] - g["dobj"]) <= 1e-8 * (1 + abs(g["dobj"]))
