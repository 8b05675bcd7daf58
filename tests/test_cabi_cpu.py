"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, keeps the reference's struct layouts and
defaults, and refuses to run without a GPU (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "abip_amd", "lib", "libabip_hip.so")):
        g.build()
    from abip_amd import _lib
    return _lib


def declared_symbols():
    names = set()
    for h in ("abip.h", "abip_hip.h", "abip_qcp.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for mm in re.finditer(r"\b(abip_[a-z_A-Z0-9]+)\s*\(", src):
            names.add(mm.group(1))
    return sorted(names)


def test_every_declared_symbol_is_exported(lib):
    L = lib.load()
    decl = declared_symbols()
    assert len(decl) >= 20
    for s in decl:
        assert hasattr(L, s), f"{s} declared in include/ but not exported"
    assert sorted(lib.EXPORTS) == decl


def test_struct_layouts_match_reference_dlong(lib):
    # sizes implied by src/abip-lp/include/abip.h with abip_int = long, abip_float = double
    assert C.sizeof(lib.ABIPMatrix) == 5 * 8
    assert C.sizeof(lib.ABIPData) == 7 * 8
    assert C.sizeof(lib.ABIPSettings) == 31 * 8
    assert C.sizeof(lib.ABIPSolution) == 3 * 8
    assert C.sizeof(lib.ABIPInfo) == 32 + 3 * 8 + 9 * 8
    assert lib.ABIPInfo.status_val.offset == 32 and lib.ABIPInfo.pobj.offset == 56
    assert lib.ABIPSettings.max_time.offset == 7 * 8 and lib.ABIPSettings.avg_criterion.offset == 30 * 8


def test_default_settings_are_the_references(lib):
    from abip_amd import default_settings
    s = default_settings()
    want = dict(max_ipm_iters=500, max_admm_iters=1000000, eps=1e-3, alpha=1.8, cg_rate=2.0, normalize=1, scale=1.0,
                rho_y=1e-3, sparsity_ratio=0.01, adaptive=1, eps_cor=0.2, eps_pen=0.1, adaptive_lookback=20,
                dynamic_x=0.8, dynamic_eta=1.1, restart_fre=1000, restart_thresh=100000, origin_rescale=0,
                pc_ruiz_rescale=1, qp_rescale=0, ruiz_iter=10, hybrid_mu=1, dynamic_sigma=-1.0, hybrid_thresh=1000.0,
                dynamic_sigma_second=0.5, half_update=0, avg_criterion=0, verbose=1, warm_start=0)  # util.c:288-329
    for k, v in want.items():
        assert getattr(s, k) == v, k


def test_version_and_linsys_switch(lib):
    L = lib.load()
    assert L.abip_version().decode().startswith("2.0.0")
    L.abip_hip_set_linsys(1)
    assert L.abip_hip_get_linsys() == 1
    L.abip_hip_set_linsys(0)
    assert L.abip_hip_get_linsys() == 0


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from abip_amd import Solver, problems
    A, b, c = problems.lp_afiro_like()
    with pytest.raises(RuntimeError):
        Solver(A, b, c, verbose=0)


def test_product_never_touches_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "abip_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"\boracle\b", txt) and not f.endswith("problems.py"):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
