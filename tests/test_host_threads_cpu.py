"""CPU: the host threads of the set-up passes (abip_amd/csrc/host_par.h) change nothing in the result.  Every threaded pass is order-preserving -- a thread
owns a range of columns, or of entries for an element-wise pass; row maxima are folded from per-thread tables; sums down a row keep their sequential pass -- so
the scaled matrix, D, E and the transposes are BIT-identical whatever the thread count.  Checked twice: the oracle-parity tests of the host code rerun with 8
threads and a grain small enough for their matrices to take the threaded paths, and one larger matrix scaled with 1 and with 8 threads."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_parity_of_the_host_code_with_threads():
    # ABIP_HIP_FORMS_ASYNC=1: the level-ordered forms of the factor built behind factor_upper's back (LdlHost::forms_job), as on large factors with a dense tail
    env = dict(os.environ, ABIP_HIP_HOST_THREADS="8", ABIP_HIP_HOST_GRAIN_DIV="100000", ABIP_HIP_FORMS_ASYNC="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_host_factor_cpu.py"),
                        os.path.join(ROOT, "tests", "test_qcp_host_cpu.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


BODY = r"""
import sys, ctypes as C
sys.path.insert(0, {root!r})
import numpy as np, scipy.sparse as sp
from abip_amd import _lib
from abip_amd.solver import default_settings
L = _lib.load()
rng = np.random.default_rng(5)
m, n = 3000, 9000
A = sp.random(m, n, density=0.004, random_state=rng, format="csc", data_rvs=lambda k: rng.uniform(-3, 3, k)) + sp.hstack([sp.identity(m), sp.csc_matrix((m, n - m))])
A = sp.csc_matrix(A); A.sort_indices()
for kw in (dict(), dict(origin_rescale=1), dict(qp_rescale=1, pc_ruiz_rescale=0)):
    Ax = A.data.astype(np.float64).copy(); Ai = A.indices.astype(np.int64); Ap = A.indptr.astype(np.int64)
    mat = _lib.ABIPMatrix(Ax.ctypes.data_as(_lib.PF), Ai.ctypes.data_as(_lib.PI), Ap.ctypes.data_as(_lib.PI), m, n)
    st = default_settings(**kw)
    D, E, means = np.zeros(m), np.zeros(n), np.zeros(2)
    assert L.abip_hip_host_normalize_A(C.byref(mat), C.byref(st), D.ctypes.data_as(_lib.PF), E.ctypes.data_as(_lib.PF), means.ctypes.data_as(_lib.PF)) == 0
    sys.stdout.buffer.write(Ax.tobytes() + D.tobytes() + E.tobytes() + means.tobytes())
"""


def test_scaling_is_bit_identical_with_one_and_with_eight_threads():
    out = []
    for threads in ("1", "8"):
        env = dict(os.environ, ABIP_HIP_HOST_THREADS=threads, ABIP_HIP_HOST_GRAIN_DIV="100")
        r = subprocess.run([sys.executable, "-c", BODY.format(root=ROOT)], env=env, cwd=ROOT, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(r.stdout)
    # the library's own chatter ("Done the ... rescaling!") is text in front of / between the payloads: identical in both runs
    assert len(out[0]) == len(out[1]) and len(out[0]) > 8 * 100000
    assert out[0] == out[1]


TINY = r"""
import sys, ctypes as C
sys.path.insert(0, {root!r})
import numpy as np, scipy.sparse as sp
from abip_amd import _lib
from abip_amd.solver import default_settings
L = _lib.load()
rng = np.random.default_rng(7)
m, n = 4, 10
A = sp.csc_matrix(sp.random(m, n, density=0.4, random_state=rng, format="csc", data_rvs=lambda k: rng.uniform(-3, 3, k)) + sp.hstack([sp.identity(m), sp.csc_matrix((m, n - m))]))
A.sort_indices()
Ax = A.data.astype(np.float64).copy(); Ai = A.indices.astype(np.int64); Ap = A.indptr.astype(np.int64)
mat = _lib.ABIPMatrix(Ax.ctypes.data_as(_lib.PF), Ai.ctypes.data_as(_lib.PI), Ap.ctypes.data_as(_lib.PI), m, n)
st = default_settings()
D, E, means = np.zeros(m), np.zeros(n), np.zeros(2)
assert L.abip_hip_host_normalize_A(C.byref(mat), C.byref(st), D.ctypes.data_as(_lib.PF), E.ctypes.data_as(_lib.PF), means.ctypes.data_as(_lib.PF)) == 0
sys.stdout.buffer.write(Ax.tobytes() + D.tobytes() + E.tobytes())
"""


def test_fewer_entries_than_threads_times_grain():
    """the row-maxima pass of the Ruiz scaling sized its per-thread tables with one thread count and ran with another when a tiny matrix met a
    divided grain (nnz / grain < threads): 4 x 10 with ~20 non-zeros, 8 threads, the grain divided down to 4 entries -- now the same bits as one thread."""
    out = []
    for threads in ("1", "8"):
        env = dict(os.environ, ABIP_HIP_HOST_THREADS=threads, ABIP_HIP_HOST_GRAIN_DIV="100000")
        r = subprocess.run([sys.executable, "-c", TINY.format(root=ROOT)], env=env, cwd=ROOT, capture_output=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
        out.append(r.stdout)
    assert out[0] == out[1] and len(out[0]) > 8 * 20
