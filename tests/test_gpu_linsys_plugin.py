"""GPU (-m gpu): the reference's linear-system plug-in interface (include/abip_linsys.h; src/abip-lp/include/linsys.h:10-91).

oracle/_ref/libabip_ref_hiplinsys.so is the reference's OWN abip.c -- its whole CPU loop, compiled from where it lies -- linked against
abip_amd/lib/libabip_hip_linsys.so in place of linsys/direct.c (oracle/Makefile, `ref`).  Run through the reference's abip_init / abip_solve, every
KKT solve, every A x / A' y and the scaling of A come from this repo; the outcome is held against the committed outputs of the reference built with
its own direct.c / indirect.c (tests/golden).  With the direct back-end the only difference is the rounding of the solve (the same iteration counts are
required); with PCG it is the order of the device's sums inside CG (counts within 3 %).  (x, y, s): 10 eps against the fixture, as for the device loop."""
import os

import numpy as np
import pytest

from _golden import info_of, load, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libabip_ref_hiplinsys.so")


@pytest.fixture(scope="module")
def po():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    if not os.path.exists(REFLIB):
        pytest.skip("oracle/_ref/libabip_ref_hiplinsys.so absent (built only where the reference tree is)")
    from oracle import pyoracle
    return pyoracle


@pytest.mark.parametrize("name,eps", [("lp_afiro_like", 1e-6), ("lp_random_sparse_small", 1e-6), ("lp_staircase", 1e-3), ("lp_multicommodity_small", 1e-4)])
@pytest.mark.parametrize("linsys", ["direct", "indirect"])
def test_reference_loop_over_the_plugin_matches_reference_fixture(po, name, eps, linsys, monkeypatch):
    monkeypatch.setenv("ABIP_HIP_LINSYS", linsys)
    z, A, b, c = load(name)
    tag = f"{linsys}_{eps:g}"
    g = info_of(z, tag)
    r = po.solve("ref", A, b, c, linsys="hiplinsys", eps=eps, verbose=0)
    assert r.info["status_val"] == g["status_val"] == 1
    assert r.info["ipm_iter"] == g["ipm_iter"]
    assert r.info["admm_iter"] == g["admm_iter"], (r.info["admm_iter"], g["admm_iter"])
    tol = 10 * eps
    for k, got in (("x", r.x), ("y", r.y), ("s", r.s)):
        assert rel(got, z[f"{tag}_{k}"]) < tol, (name, linsys, k, rel(got, z[f"{tag}_{k}"]))
    assert abs(r.info["pobj"] - g["pobj"]) <= tol * (1 + abs(g["pobj"]))
    for k in ("res_pri", "res_dual", "rel_gap"):
        assert r.info[k] < eps
