"""GPU: randomized parity sweep -- seeded random / staircase / network LPs of varied shapes, both KKT back-ends, device vs oracle
(status, iteration counts within 3 %, objective and (x, y, s) to 10 eps).  Not part of the test suite; prints one line per case."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
from abip_amd import Solver, problems
from oracle import pyoracle as po

g.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
cases = []
for t in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    kind = t % 3
    if kind == 0:
        m = int(rng.integers(20, 400)); n = int(m * rng.uniform(1.5, 4)); pc = int(rng.integers(2, 8))
        A, b, c = problems.lp_random_sparse(m=m, n=n, per_col=pc, seed=int(rng.integers(1, 10 ** 6))); tag = f"rand {m}x{n}/{pc}"
    elif kind == 1:
        st = int(rng.integers(2, 8)); rp = int(rng.integers(8, 40)); cp = int(rp * rng.uniform(1.5, 3))
        A, b, c = problems.lp_staircase(seed=int(rng.integers(1, 10 ** 6)), stages=st, rows_per=rp, cols_per=cp)[:3]; tag = f"stair {st}x{rp}x{cp}"
    else:
        nd = int(rng.integers(8, 40)); ar = int(nd * rng.uniform(2, 4)); cm = int(rng.integers(2, 6))
        A, b, c = problems.lp_multicommodity(seed=int(rng.integers(1, 10 ** 6)), nodes=nd, arcs=ar, commodities=cm)[:3]; tag = f"mc {nd}/{ar}/{cm}"
    for linsys in ("direct", "indirect"):
        eps = 1e-5
        o = po.solve("oracle", A, b, c, linsys=linsys, eps=eps, max_admm_iters=200000)
        with Solver(A, b, c, linsys=linsys, eps=eps, verbose=0, max_admm_iters=200000) as S:
            info = S.solve()
            T = int(S.scalar("tail")) if linsys == "direct" else -1
            rel = lambda a, r: np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-300)
            ex = max(rel(S.x, o.x), rel(S.y, o.y), rel(S.s, o.s)) if o.info["status_val"] == 1 else 0.0
        ok = (info["status_val"] == o.info["status_val"] and info["ipm_iter"] == o.info["ipm_iter"]
              and abs(info["admm_iter"] - o.info["admm_iter"]) <= 0.03 * o.info["admm_iter"] + 2 and ex < 10 * eps)
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {tag:22s} {linsys:8s} T={T:5d} status {info['status_val']}/{o.info['status_val']} admm {info['admm_iter']}/{o.info['admm_iter']} "
              f"ipm {info['ipm_iter']}/{o.info['ipm_iter']} rel(xys) {ex:.1e}", flush=True)
print("FAILURES", bad)
