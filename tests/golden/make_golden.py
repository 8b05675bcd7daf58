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
"""
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


def main():
    if not po.have_ref():
        po.build(ref=True)
    A, b, c = problems.lp_afiro_like()
    capture("lp_afiro_like", A, b, c, (1e-3, 1e-6, 1e-8), [1, 2, 3, 5, 10, 20, 40])
    A, b, c = problems.lp_staircase()
    capture("lp_staircase", A, b, c, (1e-3, 1e-6, 1e-8), [1, 2, 5, 10, 25])
    A, b, c = problems.lp_multicommodity(nodes=40, arcs=150, commodities=4)
    capture("lp_multicommodity_small", A, b, c, (1e-4, 1e-8), [1, 3, 10])
    A, b, c = problems.lp_random_sparse(m=300, n=800, per_col=6, seed=5)
    capture("lp_random_sparse_small", A, b, c, (1e-3, 1e-6, 1e-8), [1, 2, 5, 10])
    # non-default algorithm switches (half update, origin / qp scaling, no normalisation, no adaptive)
    A, b, c = problems.lp_random_sparse(m=60, n=150, per_col=4, seed=9)
    for tag, kw in (("half", dict(half_update=1)), ("origin", dict(origin_rescale=1, pc_ruiz_rescale=0)),
                    ("qp", dict(qp_rescale=1, pc_ruiz_rescale=0)), ("nonorm", dict(normalize=0)),
                    ("noadapt", dict(adaptive=0)), ("scale5", dict(scale=5.0)), ("tedious", dict(dynamic_sigma_second=0.0))):
        capture("lp_tiny_" + tag, A, b, c, (1e-4,), [1, 4], **kw)


if __name__ == "__main__":
    main()
