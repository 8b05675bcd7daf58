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


def test_linsys_plugin_exports_the_reference_plug_in_interface(lib):
    """lib/libabip_hip_linsys.so (include/abip_linsys.h): its dynamic symbol table is the reference's linsys.h surface as linsys/direct.c defines it
    (the nine symbols abip.c calls + free_lin_sys_work_pds) plus the back-end switch and the allocator hook -- and NOT abip_init / abip_solve, which the
    program it is linked into brings itself.  Without a GPU abip_init_lin_sys_work returns NULL (no CPU fallback)."""
    import subprocess
    path = os.path.join(ROOT, "abip_amd", "lib", "libabip_hip_linsys.so")
    if not os.path.exists(path):
        import __graft_entry__ as g
        g.build()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "abip_linsys.h")).read(), flags=re.S)
    decl = sorted({mm.group(1) for mm in re.finditer(r"\b(abip_[a-z_A-Z0-9]+)\s*\(", src)})
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if " T " in ln)
    assert exported == sorted(decl + ["abip_hip_set_linsys", "abip_hip_get_linsys"])
    L = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0))
    for s in decl:
        assert hasattr(L, s)
    import torch
    if not torch.cuda.is_available():
        import numpy as np
        from abip_amd import default_settings, problems
        A, b, c = problems.lp_afiro_like()
        Ax = np.ascontiguousarray(A.data, dtype=np.float64); Ai = np.ascontiguousarray(A.indices, dtype=np.int64); Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
        mat = lib.ABIPMatrix(Ax.ctypes.data_as(lib.PF), Ai.ctypes.data_as(lib.PI), Ap.ctypes.data_as(lib.PI), A.shape[0], A.shape[1])
        st = default_settings()
        L.abip_init_lin_sys_work.restype = C.c_void_p
        assert not L.abip_init_lin_sys_work(C.byref(mat), C.byref(st))


def test_reference_loop_links_against_the_plugin():
    """Container only: oracle/_ref/libabip_ref_hiplinsys.so is the reference's abip.c + common.c (from where they lie) linked with -z defs against the plug-in;
    every abip_* symbol it leaves undefined is one the plug-in exports."""
    import subprocess
    ref = os.path.join(ROOT, "oracle", "_ref", "libabip_ref_hiplinsys.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built (no reference tree)")
    und = [ln.split()[-1] for ln in subprocess.run(["nm", "-D", ref], check=True, capture_output=True, text=True).stdout.splitlines() if " U abip" in ln]
    assert sorted(und) == sorted(["abip_accum_by_A", "abip_accum_by_Atrans", "abip_free_lin_sys_work", "abip_get_lin_sys_method", "abip_get_lin_sys_summary",
                                  "abip_init_lin_sys_work", "abip_normalize_A", "abip_solve_lin_sys", "abip_un_normalize_A"])
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "abip_amd", "lib", "libabip_hip_linsys.so")], check=True, capture_output=True, text=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines()}
    assert set(und) <= have


def test_plugin_scaling_equals_the_references(lib):
    """abip_normalize_A of the plug-in against the reference's (linsys/common.c:150-565, through libabip_ref_direct.so): D, E and the scaled values to 1e-14,
    and un_normalize_A restores A."""
    import numpy as np
    from abip_amd import _lib, default_settings
    from _golden import load, rel
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libabip_ref_direct.so")):
        pytest.skip("oracle/_ref not built (no reference tree)")
    z, A, b, c = load("lp_multicommodity_small")
    L = C.CDLL(os.path.join(ROOT, "abip_amd", "lib", "libabip_hip_linsys.so"), mode=getattr(os, "RTLD_LOCAL", 0))
    Lr = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libabip_ref_direct.so"), mode=getattr(os, "RTLD_LOCAL", 0))

    class Scal(C.Structure):
        _fields_ = [("D", _lib.PF), ("E", _lib.PF), ("mean_norm_row_A", C.c_double), ("mean_norm_col_A", C.c_double)]

    outs = []
    for lib in (L, Lr):
        Ax = np.ascontiguousarray(A.data, dtype=np.float64).copy(); Ai = np.ascontiguousarray(A.indices, dtype=np.int64); Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
        mat = _lib.ABIPMatrix(Ax.ctypes.data_as(_lib.PF), Ai.ctypes.data_as(_lib.PI), Ap.ctypes.data_as(_lib.PI), A.shape[0], A.shape[1])
        st = default_settings()
        sc = Scal()
        lib.abip_normalize_A(C.byref(mat), C.byref(st), C.byref(sc))
        D = np.ctypeslib.as_array(sc.D, shape=(A.shape[0],)).copy(); E = np.ctypeslib.as_array(sc.E, shape=(A.shape[1],)).copy()
        scaled = Ax.copy()
        lib.abip_un_normalize_A(C.byref(mat), C.byref(st), C.byref(sc))
        outs.append((D, E, scaled, Ax.copy(), sc.mean_norm_row_A, sc.mean_norm_col_A))
    (D, E, S, back, mr, mc), (Dr, Er, Sr, backr, mrr, mcr) = outs
    assert rel(D, Dr) < 1e-14 and rel(E, Er) < 1e-14 and rel(S, Sr) < 1e-14
    assert abs(mr - mrr) <= 1e-13 * mrr and abs(mc - mcr) <= 1e-13 * mcr
    assert rel(back, A.data) < 1e-14 and rel(backr, A.data) < 1e-14


@pytest.mark.parametrize("header", ["abip.h", "abip_hip.h", "abip_qcp.h", "abip_linsys.h"])
def test_headers_compile_as_plain_c(header, tmp_path):
    """The boundary is a C ABI: every header of include/ must stand alone in a C99 translation unit (-Wall -Wextra -Werror)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text(f'#include "{header}"\nint main(void) {{ return 0; }}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)], check=True)
