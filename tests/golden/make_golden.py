"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref, compiled from
/root/reference by oracle/Makefile).  Runs only in the dev container; the fixtures it
writes are data (inputs + the reference's outputs) and are committed.

    python tests/golden/make_golden.py

Fixture layout (one .npz per instance):
    Ax, Ai, Ap, m, n, b, c                   -- the LP (CSC)
    <linsys>_<eps>_{x,y,s}                    -- final un-scaled solution (eps 1e-8 on the four main instances: the device is held to 1e-6
                                                 relative against THESE, the north-star bar, tests/test_gpu_parity.py)
    <linsys>_<eps>_info                       -- [status_val, ipm_iter, admm_iter, pobj, dobj, res_pri, res_dual, rel_gap]
    <linsys>_state_T                          -- T values at which the state was captured
    <linsys>_state_{u,v,u_t}                  -- (len(T), l) scaled iterates after exactly T inner iterations
    <linsys>_setup_{g,h,b,c}, _scal (g_th, sc_b, sc_c)   -- what setup produced

The two BASELINE-size fixtures do not store the LP -- the seeded generators of abip_amd/problems.py rebuild it, and
`lp_sha256` (over Ax, Ai, Ap, b, c) lets the tests check they rebuilt the same one:
    lp_pds_like_full.npz   -- C3 (16 390 x 48 400): the reference's full solves, <linsys>_<eps>_{x,y,s,info} at eps 1e-4 and 1e-6, both back-ends
                              (~25 min of CPU: the direct back-end runs ~10 iterations/s on this LP)
    lp_c4_prefix.npz       -- C4 (200 000 x 500 000): the reference stopped by max_admm_iters = 25 and 60 ("Solved/Inaccurate"): info, the norms and
                              sums of (x, y, s) and every STRIDE-th entry of each (the whole vectors would be 9.6 MB per run)

    python tests/golden/make_golden.py [--only lp_pds_like_full|lp_c4_prefix|small]      (ABIP_GOLDEN_CACHE=<dir>: keep / reuse the runs of the big fixtures)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from abip_amd import problems  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
INFO_KEYS = ("status_val", "ipm_iter", "admm_iter", "pobj", "dobj", "res_pri", "res_dual", "rel_gap")


def capture(name, A, b, c, eps_list, states, linsys_list=("direct", "indirect"), **kw):
    d = dict(Ax=A.data, Ai=A.indices.astype(np.int64), Ap=A.indptr.astype(np.int64), m=A.shape[0], n=A.shape[1], b=b, c=c)
    for ls in linsys_list:
        for eps in eps_list:
            r = po.solve("ref", A, b, c, linsys=ls, eps=eps, **kw)
            tag = f"{ls}_{eps:g}"
            d[tag + "_x"], d[tag + "_y"], d[tag + "_s"] = r.x, r.y, r.s
            d[tag + "_info"] = np.array([r.info[k] for k in INFO_KEYS], dtype=np.float64)
            print(name, tag, r.info["status"], r.info["ipm_iter"], r.info["admm_iter"], r.info["pobj"])
        if states:
            U, V, UT = [], [], []
            for T in states:
                r = po.state_after("ref", A, b, c, T, linsys=ls, eps=1e-9, **kw)
                U.append(r.work["u"]); V.append(r.work["v"]); UT.append(r.work["u_t"])
            d[ls + "_state_T"] = np.array(states)
            d[ls + "_state_u"], d[ls + "_state_v"], d[ls + "_state_u_t"] = np.array(U), np.array(V), np.array(UT)
            d[ls + "_setup_g"], d[ls + "_setup_h"] = r.work["g"], r.work["h"]
            d[ls + "_setup_b"], d[ls + "_setup_c"] = r.work["b"], r.work["c"]
            d[ls + "_scal"] = np.array([r.work["g_th"], r.work["sc_b"], r.work["sc_c"]])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def lp_sha256(A, b, c):
    h = hashlib.sha256()
    for a in (np.asarray(A.data, dtype=np.float64), np.asarray(A.indices, dtype=np.int64), np.asarray(A.indptr, dtype=np.int64), np.asarray(b, dtype=np.float64), np.asarray(c, dtype=np.float64)):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def ref_run(name, A, b, c, ls, tag, **kw):
    """One run of the real reference; with ABIP_GOLDEN_CACHE set its (x, y, s, info) is kept there and reused."""
    cache = os.environ.get("ABIP_GOLDEN_CACHE")
    f = os.path.join(cache, f"{name}_{tag}.npz") if cache else None
    if f and os.path.exists(f):
        z = np.load(f)
        return z["x"], z["y"], z["s"], z["info"]
    r = po.solve("ref", A, b, c, linsys=ls, **kw)
    info = np.array([r.info[k] for k in INFO_KEYS], dtype=np.float64)
    print(name, tag, r.info["status"], r.info["ipm_iter"], r.info["admm_iter"], r.info["pobj"], flush=True)
    if f:
        np.savez(f, x=r.x, y=r.y, s=r.s, info=info)
    return r.x, r.y, r.s, info


def capture_c3_full():
    """BASELINE configs[2] surrogate at full size (bench.py's c3 workload), whole solves of the reference."""
    A, b, c = problems.lp_multicommodity(nodes=1200, arcs=4400, commodities=10)
    d = dict(m=A.shape[0], n=A.shape[1], nnz=A.nnz, lp_sha256=lp_sha256(A, b, c))
    for ls in ("indirect", "direct"):
        for eps in (1e-4, 1e-6):
            tag = f"{ls}_{eps:g}"
            d[tag + "_x"], d[tag + "_y"], d[tag + "_s"], d[tag + "_info"] = ref_run("lp_pds_like_full", A, b, c, ls, tag, eps=eps)
    np.savez_compressed(os.path.join(OUT, "lp_pds_like_full.npz"), **d)


C4_STRIDE = 97


def capture_c4_prefix():
    """BASELINE configs[3] (the headline workload): the reference's first 25 and 60 ADMM iterations at eps 1e-6."""
    A, b, c = problems.lp_random_sparse()
    d = dict(m=A.shape[0], n=A.shape[1], nnz=A.nnz, lp_sha256=lp_sha256(A, b, c), stride=C4_STRIDE)
    for T in (25, 60):
        tag = f"indirect_T{T}"
        x, y, s, info = ref_run("lp_c4_prefix", A, b, c, "indirect", tag, eps=1e-6, max_admm_iters=T)
        d[tag + "_info"] = info
        for nm, v in (("x", x), ("y", y), ("s", s)):
            d[f"{tag}_{nm}_sample"] = v[::C4_STRIDE].copy()
            d[f"{tag}_{nm}_stats"] = np.array([np.linalg.norm(v), v.sum(), np.abs(v).max(), float(np.argmax(np.abs(v)))])
    np.savez_compressed(os.path.join(OUT, "lp_c4_prefix.npz"), **d)


def main():
    if not po.have_ref():
        po.build(ref=True)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None

    def want(name, small=True):
        return only is None or only == name or (small and only == "small")

    if want("lp_pds_like_full", small=False):
        capture_c3_full()
    if want("lp_c4_prefix", small=False):
        capture_c4_prefix()
    if want("lp_afiro_like"):
        A, b, c = problems.lp_afiro_like()
        capture("lp_afiro_like", A, b, c, (1e-3, 1e-6, 1e-8), [1, 2, 3, 5, 10, 20, 40])
    if want("lp_staircase"):
        A, b, c = problems.lp_staircase()
        capture("lp_staircase", A, b, c, (1e-3, 1e-6, 1e-8), [1, 2, 5, 10, 25])
    if want("lp_multicommodity_small"):
        A, b, c = problems.lp_multicommodity(nodes=40, arcs=150, commodities=4)
        capture("lp_multicommodity_small", A, b, c, (1e-4, 1e-8), [1, 3, 10])
    if want("lp_random_sparse_small"):
        A, b, c = problems.lp_random_sparse(m=300, n=800, per_col=6, seed=5)
        capture("lp_random_sparse_small", A, b, c, (1e-3, 1e-6, 1e-8), [1, 2, 5, 10])
    # non-default algorithm switches (half update, origin / qp scaling, no normalisation, no adaptive)
    A, b, c = problems.lp_random_sparse(m=60, n=150, per_col=4, seed=9)
    for tag, kw in (("half", dict(half_update=1)), ("origin", dict(origin_rescale=1, pc_ruiz_rescale=0)),
                    ("qp", dict(qp_rescale=1, pc_ruiz_rescale=0)), ("nonorm", dict(normalize=0)),
                    ("noadapt", dict(adaptive=0)), ("scale5", dict(scale=5.0)), ("tedious", dict(dynamic_sigma_second=0.0))):
        # scale5 is the knife-edge fixture of tests/test_gpu_parity.py: it also carries the reference's eps 1e-8 solutions
        if want("lp_tiny_" + tag):
            capture("lp_tiny_" + tag, A, b, c, (1e-4, 1e-8) if tag == "scale5" else (1e-4,), [1, 4], **kw)


if __name__ == "__main__":
    main()
