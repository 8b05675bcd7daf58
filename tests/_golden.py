"""Helpers to read the committed reference fixtures (tests/golden/*.npz)."""
import os

import numpy as np
import scipy.sparse as sp

INFO_KEYS = ("status_val", "ipm_iter", "admm_iter", "pobj", "dobj", "res_pri", "res_dual", "rel_gap")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY_VARIANTS = {
    "half": dict(half_update=1), "origin": dict(origin_rescale=1, pc_ruiz_rescale=0),
    "qp": dict(qp_rescale=1, pc_ruiz_rescale=0), "nonorm": dict(normalize=0), "noadapt": dict(adaptive=0),
    "scale5": dict(scale=5.0), "tedious": dict(dynamic_sigma_second=0.0),
}


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    A = sp.csc_matrix((z["Ax"], z["Ai"], z["Ap"]), shape=(int(z["m"]), int(z["n"])))
    return z, A, z["b"], z["c"]


def info_of(z, tag):
    return dict(zip(INFO_KEYS, z[tag + "_info"]))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def lp_sha256(A, b, c):
    """Checksum of an LP as tests/golden/make_golden.py took it: the BASELINE-size fixtures store it instead of the LP (the seeded generator rebuilds it)."""
    import hashlib
    h = hashlib.sha256()
    for a in (np.asarray(A.data, dtype=np.float64), np.asarray(A.indices, dtype=np.int64), np.asarray(A.indptr, dtype=np.int64), np.asarray(b, dtype=np.float64), np.asarray(c, dtype=np.float64)):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
