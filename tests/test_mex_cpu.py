"""CPU: the Matlab gateway source (mex/abip_hip_mex.c) compiles against include/abip.h for both back-ends (mock mex.h: no
Matlab here), reads exactly the field names the reference's gateway reads, and returns the reference's info fields."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gateway_compiles_for_both_back_ends(tmp_path):
    for extra in ([], ["-DABIP_HIP_PCG"]):
        out = tmp_path / ("gw%d.o" % len(extra))
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-c", os.path.join(ROOT, "mex", "abip_hip_mex.c"), "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "tests", "mock_mex"), "-o", str(out)] + extra, check=True)
        syms = subprocess.run(["nm", str(out)], capture_output=True, text=True, check=True).stdout
        assert " T mexFunction" in syms and " U abip_main" in syms and " U abip_hip_set_linsys" in syms


def test_gateway_field_names_are_the_references():
    src = open(os.path.join(ROOT, "mex", "abip_hip_mex.c")).read()
    from abip_amd.api import _MEX_FIELDS
    read = set(re.findall(r'_FIELD\("(\w+)"', src)) | set(re.findall(r'get_field_or\(settings, "(\w+)"', src))
    assert read == set(_MEX_FIELDS)                       # the Python mirror and the C gateway read the same names
    for f in ("status", "ipm_iter", "admm_iter", "mu", "pobj", "dobj", "resPri", "resDual", "relGap", "resInfeas", "resUnbdd", "setupTime", "solveTime"):
        assert '"%s"' % f in src                           # abip_mex.c:101-102
