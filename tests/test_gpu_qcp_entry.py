"""GPU (-m gpu): the conic entry under the reference's own names.  libabip_hip_qcp.so exports `abip` and `abip_set_default_settings` with the
reference's conic struct layouts (src/abip-qcp/include/abip.h:63-241); oracle/_ref/qcp_toy_refheader is tests/c/qcp_toy_refheader.c compiled -- in the
container that holds the reference -- against the reference's OWN header and linked against that library.  It must reproduce the reference's
recorded output on its literal toy problem (test/test_abip_install.m:32-43)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = os.path.join(ROOT, "oracle", "_ref", "qcp_toy_refheader")


@pytest.mark.gpu
def test_reference_header_program_reproduces_the_recorded_toy_output():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(PROG), "oracle/_ref/qcp_toy_refheader is built by `make -C oracle ref` where /root/reference exists and travels with the snapshot"
    out = subprocess.run([PROG], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    m = re.search(r"status (-?\d+) ipm (\d+) admm (\d+) pobj (\S+) dobj (\S+) x0 (\S+)", out.stdout)
    assert m, out.stdout
    assert int(m.group(1)) == 1 and int(m.group(2)) == 10 and int(m.group(3)) == 91
    assert abs(float(m.group(4)) - (-0.984063813)) < 5e-9 and abs(float(m.group(5)) - (-0.984063938)) < 5e-9


def test_conic_entry_library_exports_exactly_the_reference_names():
    lib = os.path.join(ROOT, "abip_amd", "lib", "libabip_hip_qcp.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout.split("\n")
    names = sorted(l.split()[-1] for l in syms if l.strip())
    assert names == ["abip", "abip_set_default_settings"]
