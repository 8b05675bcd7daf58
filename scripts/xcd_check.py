"""Developer check of the one-XCD persistent launch (dev_xcd.h): the same LP with ABIP_HIP_XCD=1 and =0, iteration counts, solutions, rates."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abip_amd import problems
from abip_amd.solver import Solver

def run(A, b, c, linsys, xcd, eps, verbose=0, max_steps=None):
    os.environ["ABIP_HIP_XCD"] = "1" if xcd else "0"
    os.environ["ABIP_HIP_XCD_OUTER"] = "1" if xcd == 2 else "0"   # 2: launches that span outer iterations (round 4); 1: one batch of inner iterations per launch
    with Solver(A, b, c, linsys=linsys, eps=eps, verbose=verbose) as s:
        on = s.scalar("xcd")
        t0 = time.time()
        out = s.solve()
        dt = time.time() - t0
        return dict(xcd=on, g=s.scalar("xcd_g"), nz=s.scalar("xcd_nz"), batches=s.scalar("xcd_batches"), exch=s.scalar("xcd_exchanges"), launches=s.scalar("xcd_launches"),
                    outer=s.scalar("xcd_outer_done"), look=s.scalar("xcd_lookaheads"), giveups=s.scalar("xcd_giveups"), dt=dt, x=s.x.copy(), y=s.y.copy(), s=s.s.copy(), **out)

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    linsys = sys.argv[2] if len(sys.argv) > 2 else "indirect"
    eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
    if which == "small":
        A, b, c = problems.lp_multicommodity(nodes=40, arcs=160, commodities=4)
    elif which == "c3":
        A, b, c = problems.lp_multicommodity(nodes=1200, arcs=4400, commodities=10)
    elif which == "c2":
        A, b, c = problems.lp_staircase()
    elif which == "afiro":
        A, b, c = problems.lp_afiro_like()
    elif which.startswith("rand:"):      # rand:m:n:per_col
        _, m_, n_, pc = which.split(":")
        A, b, c = problems.lp_random_sparse(m=int(m_), n=int(n_), per_col=int(pc), seed=3)
    elif which.startswith("stair:"):     # stair:stages:rows_per:cols_per
        _, st_, rp, cp = which.split(":")
        A, b, c = problems.lp_staircase(stages=int(st_), rows_per=int(rp), cols_per=int(cp))[:3]
    elif which.startswith("mc:"):        # mc:nodes:arcs:commodities
        _, nd, ar, cm = which.split(":")
        A, b, c = problems.lp_multicommodity(nodes=int(nd), arcs=int(ar), commodities=int(cm))
    print("problem", which, A.shape, A.nnz, "linsys", linsys, "eps", eps, flush=True)
    res = {}
    verbose = int(os.environ.get("XCD_CHECK_VERBOSE", "0"))
    for xcd in (2, 1, 0):
        r = run(A, b, c, linsys, xcd, eps, verbose=verbose)
        res[xcd] = r
        print("xcd=%d on=%g G=%g nz=%g launches=%g (outer iterations inside: %g, look-aheads %g, give-ups %g) exch=%g status=%s admm=%d ipm=%d pobj=%.10e time=%.3fs  -> %.0f it/s" % (
            xcd, r["xcd"], r["g"], r["nz"], r["launches"], r["outer"], r["look"], r["giveups"], r["exch"], r["status"], r["admm_iter"], r["ipm_iter"], r["pobj"], r["dt"], r["admm_iter"] / r["dt"]), flush=True)
    for u, v in ((2, 0), (1, 0), (2, 1)):
        a, b_ = res[u], res[v]
        print("xcd=%d against xcd=%d:" % (u, v), " ".join("rel diff %s %.3e" % (k, np.linalg.norm(a[k] - b_[k]) / max(1e-300, np.linalg.norm(b_[k]))) for k in ("x", "y", "s")))

if __name__ == "__main__":
    main()
