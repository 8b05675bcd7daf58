"""Where the host half of the direct back-end's set-up goes on C5's matrix, without a GPU (and without the host's dense LDL' of the tail):
    ABIP_HIP_SETUP_TIMES=1 python scripts/host_setup_time.py [reps]
(the conic path brings its own elimination order; this LP-form KKT of the same A runs the minimum-degree pass instead -- ignore the "ordering" line)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["ABIP_HIP_HOST_FACTOR_ONLY"] = "1"
from abip_amd import problems  # noqa: E402
from test_host_factor_cpu import host_solve  # noqa: E402

data, K = problems.qcp_lasso_socp()
A = data["A"]
print(A.shape, A.nnz, flush=True)
rhs = np.ones(A.shape[0] + A.shape[1])
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t = time.time(); z, st = host_solve(A, 1.0, -1, rhs); print("total %.3f s" % (time.time() - t), st, flush=True)
