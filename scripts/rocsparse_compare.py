"""GPU box, calibration only: rocSPARSE's CSR SpMV on the SCALED C4 matrix -- A (m rows, gathers the n-vector: what k_cg_spmv_A multiplies)
and A' (n rows: k_cg_spmv_At) -- next to this repository's kernels on the same arrays.  Writes the two matrices to /tmp and runs tools/rocsparse_probe on them."""
import os, struct, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from abip_amd import problems
from abip_amd.solver import Solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A, b, c = problems.lp_random_sparse()
m, n = A.shape
with Solver(A, b, c, linsys="indirect", eps=1e-6, verbose=0) as S:
    Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)     # the scaled matrix the kernels multiply
    S.profile_enable_stamps(("spmv_At", "spmv_A"))
    S.begin(); S.step(30); S.sync()
    p = S.profile_read()
    ours = {k: 1e3 * p["stamp_ms"][k] / max(p["stamp_launches"][k], 1) for k in ("spmv_A", "spmv_At")}
def dump(M, path):
    M = sp.csr_matrix(M); M.sort_indices()
    with open(path, "wb") as f:
        f.write(struct.pack("qqq", M.shape[0], M.shape[1], M.nnz))
        f.write(M.indptr.astype(np.int32).tobytes()); f.write(M.indices.astype(np.int32).tobytes()); f.write(M.data.astype(np.float64).tobytes())
dump(Asc, "/tmp/c4_A.bin"); dump(Asc.T, "/tmp/c4_At.bin")
print("this repository, device-side stamps of the launches that did work (30 ADMM steps): k_cg_spmv_A %.2f us, k_cg_spmv_At %.2f us" % (ours["spmv_A"], ours["spmv_At"]), flush=True)
for f in ("/tmp/c4_A.bin", "/tmp/c4_At.bin"):
    subprocess.run([os.path.join(ROOT, "tools", "rocsparse_probe"), f], check=False)
