"""CPU (build container only, skipped where /root/reference is absent): the reference's own Matlab gateway
src/abip-lp/mexfile/abip_mex.c compiles from where it lies with the REFERENCE's headers (make_abip.m's flags) and every abip_*
symbol it needs is exported by libabip_hip.so -- the binding INTEGRATION.md section 1 describes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/abip-lp"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent")
def test_reference_gateway_links_against_the_library(tmp_path):
    import __graft_entry__ as g
    g.build()
    obj = tmp_path / "abip_mex.o"
    subprocess.run(["gcc", "-O2", "-w", "-fPIC", "-c", "-DDLONG", "-DCOPYAMATRIX", "-DMATLAB_MEX_FILE", "-I", os.path.join(ROOT, "tests", "mock_mex"),
                    "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "linsys"), os.path.join(REF, "mexfile", "abip_mex.c"), "-o", str(obj)], check=True)
    und = {ln.split()[-1] for ln in subprocess.run(["nm", "--undefined-only", str(obj)], capture_output=True, text=True, check=True).stdout.splitlines()}
    need = {s for s in und if s.startswith("abip_")}
    assert need == {"abip_main", "abip_set_default_settings"}
    exp = {ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "abip_amd", "lib", "libabip_hip.so")],
                                                    capture_output=True, text=True, check=True).stdout.splitlines()}
    assert need <= exp
    assert all(s.startswith("abip_") for s in exp), sorted(s for s in exp if not s.startswith("abip_"))   # -fvisibility=hidden + version script: the C ABI and nothing else
    so = os.path.join(ROOT, "oracle", "_ref", "libmexgw_ref.so")
    assert os.path.exists(so)                                            # built by `make -C oracle ref` (build())
    ldd = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "libabip_hip.so" in ldd and "not found" not in ldd


def test_int32_library_exports_the_same_abi():
    exp = []
    for nm_ in ("libabip_hip.so", "libabip_hip32.so"):
        p = os.path.join(ROOT, "abip_amd", "lib", nm_)
        if not os.path.exists(p):
            import __graft_entry__ as g
            g.build()
        exp.append({ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--defined-only", p], capture_output=True, text=True, check=True).stdout.splitlines()})
    assert exp[0] == exp[1] and "abip_main" in exp[0]


def test_conic_struct_layouts_equal_the_reference_header(tmp_path):
    """include/abip_qcp.h against the reference's OWN conic header (src/abip-qcp/include/abip.h:63-241, which needs no MKL): sizes and offsets of
    ABIPData / ABIPSettings / ABIPCone / ABIPSolution / ABIPInfo, as the compiler lays them out (the reference builds its conic code with abip_int = int:
    glbopts.h:10 leaves DLONG commented out).  Container only (the reference tree is not on the GPU box)."""
    import os, subprocess
    ref = "/root/reference/src/abip-qcp/include"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree is not here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = """
    #include <stdio.h>
    #include <stddef.h>
    #include "%s"
    #define P(T, f) printf(#f " %%zu\\n", offsetof(T, f))
    int main(void) {
      printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(%sData), sizeof(%sSettings), sizeof(%sCone), sizeof(%sSolution), sizeof(%sInfo), sizeof(%sMatrix));
      P(%sData, A); P(%sData, Q); P(%sData, b); P(%sData, c); P(%sData, lambda); P(%sData, stgs);
      P(%sSettings, rho_y); P(%sSettings, eps_unb); P(%sSettings, alpha); P(%sSettings, verbose); P(%sSettings, linsys_solver); P(%sSettings, prob_type);
      P(%sSettings, time_limit); P(%sSettings, psi); P(%sSettings, pc_scaling);
      P(%sCone, q); P(%sCone, qsize); P(%sCone, rq); P(%sCone, rqsize); P(%sCone, f); P(%sCone, z); P(%sCone, l);
      P(%sInfo, status_val); P(%sInfo, admm_iter); P(%sInfo, pobj); P(%sInfo, rel_gap); P(%sInfo, solve_time); P(%sInfo, avg_cg_iters);
      return 0;
    }
    """
    outs = []
    for hdr, pre, inc in (("abip.h", "ABIP", ref), ("abip_qcp.h", "QCP", os.path.join(root, "include"))):
        src = tmp_path / f"lay_{pre}.c"
        src.write_text(body % ((hdr,) + (pre,) * 34))
        exe = tmp_path / f"lay_{pre}"
        subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe)], check=True)
        outs.append(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)
    assert outs[0] == outs[1] and len(outs[0].split("\n")) > 25
