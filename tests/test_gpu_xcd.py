"""GPU (-m gpu): the one-XCD persistent launch (abip_amd/csrc/dev_xcd.h) -- the whole inner ADMM loop of a cache-resident LP in one kernel on
the 32 CUs of one XCD (or of 2, 4, 8 XCDs) -- against the REFERENCE's eps = 1e-8 fixtures (tests/golden, written from oracle/_ref) where a fixture exists, and
against the launch path (one kernel per step of the iteration, dev_kernels.h; held to the same fixtures by test_gpu_parity.py) on generated LPs.  (With the
launch on by default the fixture / oracle tests of test_gpu_parity.py run through it too.)

Bars: the reference's outer AND inner iteration counts, (x, y, s) within 1e-6 relative at eps = 1e-8 (the north-star bar: the paths add in different orders and
the direct variant applies inv(rho I + A A') where the reference solves with LDL'); bit-identical results whatever the batching of the iterations.
(Round 4 compared the persistent launch with a launch-path run of the same LP; the fixture is the stronger bar and costs no second solve.)"""
import numpy as np
import pytest

from _golden import info_of, load, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    import abip_amd
    return abip_amd


def _against_the_reference_at_1e8(S, info, z, linsys):
    """The reference's own eps = 1e-8 run of this LP: its counts, its (x, y, s) to 1e-6, its objective."""
    tag = f"{linsys}_1e-08"
    g = info_of(z, tag)
    assert info["status_val"] == g["status_val"] == 1
    assert (info["ipm_iter"], info["admm_iter"]) == (g["ipm_iter"], g["admm_iter"]), (info["ipm_iter"], info["admm_iter"], g["ipm_iter"], g["admm_iter"])
    assert abs(info["pobj"] - g["pobj"]) <= 1e-6 * (1 + abs(g["pobj"]))
    for k in "xys":
        assert rel(getattr(S, k), z[f"{tag}_{k}"]) < 1e-6, k


@pytest.mark.parametrize("name", ["lp_afiro_like", "lp_random_sparse_small", "lp_multicommodity_small", "lp_staircase"])
@pytest.mark.parametrize("linsys", ["indirect", "direct"])
def test_one_xcd_launch_matches_the_reference_at_tight_eps(gpu, name, linsys, monkeypatch):
    z, A, b, c = load(name)
    monkeypatch.setenv("ABIP_HIP_XCD", "1")
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-8) as S:
        assert S.scalar("xcd") == 1.0
        info = S.solve()
        assert S.scalar("xcd_launches") > 0
        _against_the_reference_at_1e8(S, info, z, linsys)


@pytest.mark.parametrize("name,linsys", [("lp_multicommodity_small", "indirect"), ("lp_staircase", "direct"), ("lp_afiro_like", "indirect")])
def test_one_xcd_launch_is_bit_identical_whatever_the_batching(gpu, name, linsys, monkeypatch):
    """One launch per inner loop, one launch per iteration (ABIP_HIP_BATCH=0), or strides of 7 through the stepping ABI: the same bits.  Every
    workgroup adds the partial sums of an exchange in rank order, so neither the batching nor the workgroup-to-CU placement can change a sum."""
    z, A, b, c = load(name)
    monkeypatch.setenv("ABIP_HIP_XCD", "1")
    runs = []
    for mode in ("batched", "stepwise", "strided"):
        if mode == "stepwise":
            monkeypatch.setenv("ABIP_HIP_BATCH", "0")
        else:
            monkeypatch.delenv("ABIP_HIP_BATCH", raising=False)
        with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-5) as S:
            assert S.scalar("xcd") == 1.0
            if mode == "strided":
                S.begin()
                fin = False
                while not fin:
                    fin, done = S.step(7)
                    assert done <= 7
                info = S.end()
            else:
                info = S.solve()
            runs.append((info["admm_iter"], info["ipm_iter"], info["pobj"], S.scalar("tot_cg_its"), S.x.copy(), S.y.copy(), S.s.copy()))
    for r in runs[1:]:
        assert r[:4] == runs[0][:4]
        for a2, b2 in zip(r[4:], runs[0][4:]):
            assert np.array_equal(a2, b2)


def test_one_xcd_launch_run_to_run_identical(gpu, monkeypatch):
    """Two solves of the same LP in one process (different tickets, tags continuing): identical bits."""
    z, A, b, c = load("lp_multicommodity_small")
    monkeypatch.setenv("ABIP_HIP_XCD", "1")
    res = []
    for _ in range(2):
        with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-6) as S:
            info = S.solve()
            res.append((info["admm_iter"], info["pobj"], S.x.copy(), S.y.copy(), S.s.copy()))
    assert res[0][:2] == res[1][:2]
    for a2, b2 in zip(res[0][2:], res[1][2:]):
        assert np.array_equal(a2, b2)


def test_problems_that_do_not_fit_stay_on_the_launch_path(gpu, monkeypatch):
    """The persistent launch takes an LP only when its slices fit the registers / LDS of the workgroups of the 8 XCDs with at most 8 non-zeros per
    thread (~8e5 non-zeros; beyond, the launch path) and, for the direct back-end, when the dense inverse of the m x m Schur complement
    is affordable (m <= 6144): otherwise abip_init leaves the launch path in charge, silently.  (The plan itself: tests/test_xcd_plan_cpu.py.)"""
    from abip_amd import problems
    monkeypatch.setenv("ABIP_HIP_XCD", "1")
    A, b, c = problems.lp_random_sparse(m=20000, n=50000, per_col=32, seed=3)      # 9.8e5 non-zeros: more than 8 per thread on 256 workgroups
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-3, max_admm_iters=20) as S:
        assert S.scalar("xcd") == 0.0
        S.solve()
    A, b, c = problems.lp_random_sparse(m=6500, n=13000, per_col=4, seed=4)        # direct: m > 6144
    with gpu.Solver(A, b, c, linsys="direct", verbose=0, eps=1e-3, max_admm_iters=20) as S:
        assert S.scalar("xcd") == 0.0
        S.solve()
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-3, max_admm_iters=20) as S:
        assert S.scalar("xcd") == 1.0
        info = S.solve()
        assert info["admm_iter"] >= 20


# (BASELINE configs[1] / configs[2] at full size on both paths: tests/test_gpu_baseline_size.py and the lp_staircase fixture tests -- against the reference.)


@pytest.mark.parametrize("per_col", [16, 24])
def test_the_largest_class_the_launch_takes_agrees_with_the_launch_path(gpu, per_col, monkeypatch):
    """5e5 / 7.4e5 non-zeros: six / eight per thread on the 256 workgroups of all eight XCDs (until round 5 the launch path's).  No reference run at this size fits the
    suite; the launch path is the pinned one (fixtures, C3 / C4 at BASELINE size): same outer and inner counts, (x, y, s) to 1e-6."""
    from abip_amd import problems
    A, b, c = problems.lp_random_sparse(m=20000, n=50000, per_col=per_col, seed=3)[:3]
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ABIP_HIP_XCD", mode)
        with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=1e-4) as S:
            assert S.scalar("xcd") == float(mode)
            if mode == "1":
                assert S.scalar("xcd_g") == 256.0
            info = S.solve()
            out[mode] = (info, S.x.copy(), S.y.copy(), S.s.copy())
    a, l = out["1"], out["0"]
    assert a[0]["status_val"] == l[0]["status_val"] == 1
    assert (a[0]["ipm_iter"], a[0]["admm_iter"]) == (l[0]["ipm_iter"], l[0]["admm_iter"]), (a[0]["ipm_iter"], a[0]["admm_iter"], l[0]["ipm_iter"], l[0]["admm_iter"])
    for k in (1, 2, 3):
        assert rel(a[k], l[k]) < 1e-6


@pytest.mark.parametrize("G", [64, 128, 256])
@pytest.mark.parametrize("name,linsys", [("lp_multicommodity_small", "indirect"), ("lp_random_sparse_small", "indirect"), ("lp_staircase", "direct"),
                                         ("lp_random_sparse_small", "direct")])
def test_launch_spread_over_several_xcds_matches_the_reference(gpu, name, linsys, G, monkeypatch):
    """The workgroups may sit on 2, 4 or 8 XCDs (G = 64, 128, 256; chosen from the non-zero count -- PCG -- or from m -- direct --, here forced): the
    stores of an exchange are then written through (the L2s of two XCDs are not coherent with each other).  Same bar as on one XCD -- the reference's eps 1e-8
    fixture, counts included -- and the same bits whatever the batching: the partial sums are added in rank order, 64 ranks at a time."""
    z, A, b, c = load(name)
    monkeypatch.setenv("ABIP_HIP_XCD_G", str(G))
    monkeypatch.setenv("ABIP_HIP_XCD", "1")
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-8) as S:
        assert S.scalar("xcd") == 1.0 and S.scalar("xcd_g") == float(G)
        info = S.solve()
        _against_the_reference_at_1e8(S, info, z, linsys)
    runs = []
    for batch in ("1", "0"):
        monkeypatch.setenv("ABIP_HIP_BATCH", batch)
        with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-5) as S:
            info = S.solve()
            runs.append((info["admm_iter"], S.x.copy(), S.y.copy()))
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])


def test_direct_variant_beyond_m_4096_agrees_with_the_launch_path(gpu, monkeypatch):
    """m = 5000 (round 4's limit was 4096): rows of the dense inverse streamed from the L2 / HBM on eight XCDs; same counts as the launch path, (x, y, s) to 1e-6."""
    from abip_amd import problems
    A, b, c = problems.lp_staircase(stages=50, rows_per=100, cols_per=230)[:3]
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ABIP_HIP_XCD", mode)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0, eps=1e-6) as S:
            assert S.scalar("xcd") == float(mode)
            info = S.solve()
            out[mode] = (info, S.x.copy(), S.y.copy(), S.s.copy())
    a, l = out["1"], out["0"]
    assert a[0]["status_val"] == l[0]["status_val"] == 1
    assert (a[0]["ipm_iter"], a[0]["admm_iter"]) == (l[0]["ipm_iter"], l[0]["admm_iter"]), (a[0]["ipm_iter"], a[0]["admm_iter"], l[0]["ipm_iter"], l[0]["admm_iter"])
    for k in (1, 2, 3):
        assert rel(a[k], l[k]) < 1e-6


@pytest.mark.parametrize("m", [63, 65, 129, 1000, 1024, 1025, 1100, 1300, 1600])
def test_direct_variant_row_counts_around_its_paddings(gpu, m, monkeypatch):
    """The dense product of the direct variant reads its rows without a guard per element: out to 64 x 16 columns on small systems (m_pad <= 1024: the pad
    behind the last LDS row is zero, the lanes' entries of w are zero beyond m), four columns per lane at a time on larger ones (LDS rows) or eight (rows
    streamed from the L2), w zero-padded to m_pad.  Row counts one short of, on, and one past a multiple of 64, the last small size (1024), the first sizes of
    the larger form on one XCD (1025, 1100: forced -- the planner gives them four), on four (1300) and on eight (1600): the persistent launch agrees with the launch
    path as on the fixtures."""
    from abip_amd import problems
    A, b, c = problems.lp_random_sparse(m=m, n=int(2.3 * m) + 7, per_col=4, seed=1000 + m)[:3]
    out = {}
    if m in (1025, 1100):
        monkeypatch.setenv("ABIP_HIP_XCD_G", "32")
    for mode in ("1", "0"):
        monkeypatch.setenv("ABIP_HIP_XCD", mode)
        with gpu.Solver(A, b, c, linsys="direct", verbose=0, eps=1e-8) as S:
            assert S.scalar("xcd") == float(mode)
            if mode == "1":
                assert S.scalar("xcd_g") == (32.0 if m <= 1100 else 128.0 if m <= 1500 else 256.0)
            info = S.solve()
            out[mode] = (info, S.x.copy(), S.y.copy(), S.s.copy())
    a, l = out["1"], out["0"]
    assert a[0]["status_val"] == l[0]["status_val"] == 1 and a[0]["ipm_iter"] == l[0]["ipm_iter"]
    assert a[0]["admm_iter"] == l[0]["admm_iter"], (a[0]["admm_iter"], l[0]["admm_iter"])
    for k in (1, 2, 3):
        assert rel(a[k], l[k]) < 1e-6
