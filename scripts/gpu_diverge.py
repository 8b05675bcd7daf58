"""GPU: where does a device trajectory leave the oracle's?  Replays the case list of gpu_sweep.py (same seed), picks case #idx,
steps both and reports the first iteration whose (u, v, u_t) differ by more than 1e-9, with the CG counts around it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
from abip_amd import Solver, problems
from oracle import pyoracle as po
g.build()
seed, idx, linsys = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(seed)
for t in range(idx + 1):
    kind = t % 3
    if kind == 0:
        m = int(rng.integers(20, 400)); n = int(m * rng.uniform(1.5, 4)); pc = int(rng.integers(2, 8))
        args = dict(m=m, n=n, per_col=pc, seed=int(rng.integers(1, 10 ** 6))); gen = problems.lp_random_sparse
    elif kind == 1:
        st = int(rng.integers(2, 8)); rp = int(rng.integers(8, 40)); cp = int(rp * rng.uniform(1.5, 3))
        args = dict(seed=int(rng.integers(1, 10 ** 6)), stages=st, rows_per=rp, cols_per=cp); gen = problems.lp_staircase
    else:
        nd = int(rng.integers(8, 40)); ar = int(nd * rng.uniform(2, 4)); cm = int(rng.integers(2, 6))
        args = dict(seed=int(rng.integers(1, 10 ** 6)), nodes=nd, arcs=ar, commodities=cm); gen = problems.lp_multicommodity
A, b, c = gen(**args)[:3]
print(gen.__name__, args, A.shape)
T = int(sys.argv[4]) if len(sys.argv) > 4 else 7000
o = po.solve("oracle", A, b, c, linsys=linsys, eps=1e-5, trace=T, max_admm_iters=200000)
print("oracle", o.info["admm_iter"], o.info["ipm_iter"], o.info["pobj"], o.info["dobj"])
rel = lambda a, r: np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-300)
with Solver(A, b, c, linsys=linsys, eps=1e-5, verbose=0, max_admm_iters=200000) as S:
    S.begin()
    first = None
    hist = []
    for t in range(min(T, len(o.trace))):
        cg0 = S.scalar("tot_cg_its")
        fin, done = S.step(1)
        e = max(rel(S.vector(nm), o.trace[t, col]) for col, nm in enumerate(("u", "v", "u_t")))
        hist.append((t + 1, e, S.scalar("tot_cg_its") - cg0, S.scalar("mu"), S.scalar("ipm_iter")))
        if e > 1e-9 and first is None:
            first = t + 1
        if fin or (first is not None and t + 1 > first + 3):
            break
    print("first divergence at iteration", first)
    for h in hist[-12:]:
        print("  it %d rel %.2e cg %d mu %.3e outer %d" % h)
    info = S.end()
print("device (stopped early)", info["pobj"], info["dobj"])
