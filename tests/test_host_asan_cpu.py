"""CPU: the host set-up code (ordering, factorisation, level sets) under AddressSanitizer + UBSan.  The GPU pool offers no device
sanitizer, so the host half -- where the index gymnastics live -- is checked here with g++ -fsanitize on a spread of matrix shapes."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from abip_amd import problems
from test_host_factor_cpu import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def asan_tool(tmp_path_factory):
    out = tmp_path_factory.mktemp("asan") / "ldl_stats_asan"
    cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "abip_amd", "csrc"), os.path.join(ROOT, "tools", "ldl_stats.cpp"), os.path.join(ROOT, "abip_amd", "csrc", "host_setup.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime here: " + r.stderr[-300:])
    return str(out)


def dump(path, A):
    A = sp.csc_matrix(A); A.sort_indices()
    with open(path, "wb") as f:
        np.array([A.shape[0], A.shape[1], A.nnz], dtype=np.int64).tofile(f)
        A.indptr.astype(np.int64).tofile(f); A.indices.astype(np.int64).tofile(f); A.data.astype(np.float64).tofile(f)


@pytest.mark.parametrize("tail", ["auto", "0", "64"])
@pytest.mark.parametrize("name", sorted(CASES) + ["random_mid"])
def test_host_setup_is_clean_under_sanitizers(asan_tool, tmp_path, name, tail):
    A = problems.lp_random_sparse(m=900, n=2500, per_col=4, seed=3)[0] if name == "random_mid" else CASES[name]()
    path = str(tmp_path / "m.bin")
    dump(path, A)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    if tail != "auto":
        env["ABIP_HIP_TAIL"] = tail
    else:
        env.pop("ABIP_HIP_TAIL", None)
    r = subprocess.run([asan_tool, path], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-1500:]
    assert "N %d" % (A.shape[0] + A.shape[1]) in r.stdout
