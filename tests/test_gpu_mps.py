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
