"""GPU: the front end of SURVEY 8(f1) end to end -- MPS file -> standard form (preprocess.m rules) -> abip(data, K, params) on the
device, both KKT back-ends -- against HiGHS on the original bounded problem."""
import os

import numpy as np
import pytest

from test_mps_cpu import DATA, bounded_lp, highs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    import abip_amd
    return abip_amd


@pytest.mark.parametrize("pcg", [0, 1])
@pytest.mark.parametrize("case", ["testprob", "generated"])
def test_mps_file_to_device_solution(gpu, tmp_path, case, pcg):
    from abip_amd import mps
    if case == "testprob":
        path = os.path.join(DATA, "testprob.mps")
    else:
        path = str(tmp_path / "gen.mps")
        mps.mpswrite(path, bounded_lp(seed=4, n=60, me=8, mi=14))
    prob = mps.mpsread(path)
    want = highs(prob)
    A, b, c, data = mps.load_standard_form(path)
    p = gpu.abip_get_params(); p["pcg"] = pcg; p["tol"] = 1e-7; p["verbose"] = 0
    x, y, s, info = gpu.abip(dict(A=A, b=b, c=c), dict(l=A.shape[1]), p)
    assert info["status"] == "Solved"
    assert abs(info["pobj"] + data["objcon"] - (want.fun + prob["objcon"])) <= 1e-5 * (1 + abs(want.fun))
    xo = x[: data["n_orig"]] + data["lb_shift"]
    assert np.linalg.norm(xo - want.x) <= 1e-3 * (1 + np.linalg.norm(want.x))      # (the LP optimum is unique for these data)
    assert np.all(xo >= prob["lb"] - 1e-6) and np.all(xo <= prob["ub"] + 1e-6)


@pytest.mark.parametrize("xcd", ["1", "0"])
@pytest.mark.parametrize("linsys", ["direct", "indirect"])
def test_netlib_afiro_on_the_device(linsys, xcd, monkeypatch):
    """BASELINE configs[0] on the real Netlib file: published optimum -4.6475314286e+02 on both back-ends, through the one-XCD launch and the launch path."""
    import os
    import numpy as np
    import abip_amd
    from abip_amd import mps
    monkeypatch.setenv("ABIP_HIP_XCD", xcd)
    A, b, c, extra = mps.load_standard_form(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "afiro.mps"))
    with abip_amd.Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-8) as S:
        assert S.scalar("xcd") == float(xcd)
        info = S.solve()
        assert info["status"] == "Solved"
        assert abs(info["pobj"] - (-464.7531428571)) <= 1e-6 * 465 and abs(info["dobj"] - (-464.7531428571)) <= 1e-6 * 465
        x = S.x[:32]
        assert abs(float(c[:32] @ x) - (-464.7531428571)) <= 1e-6 * 465
