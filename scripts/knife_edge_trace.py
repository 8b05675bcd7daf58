"""Where does the device leave the reference's trajectory on the knife-edge fixture (lp_tiny_scale5, eps 1e-8)?  Both are run under max_ipm_iters = 1, 2, ...: after i outer
iterations the oracle (bit-pinned on the reference) and the device report (admm_iter, mu, beta); the first row that differs names the outer iteration whose
Barzilai-Borwein search (or exit test) fell on the other side.    python scripts/knife_edge_trace.py [direct|indirect] [eps] [xcd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from _golden import TINY_VARIANTS, load
linsys = sys.argv[1] if len(sys.argv) > 1 else "direct"
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
os.environ["ABIP_HIP_XCD"] = sys.argv[3] if len(sys.argv) > 3 else "0"
import abip_amd as gpu
from oracle import pyoracle as po
z, A, b, c = load("lp_tiny_scale5")
kw = TINY_VARIANTS["scale5"]
print("%3s | %-44s | %-44s" % ("i", "oracle: admm_iter  mu  beta", "device: admm_iter  mu  beta"))
for i in range(1, 18):
    o = po.solve("oracle", A, b, c, linsys=linsys, eps=eps, max_ipm_iters=i, **kw)
    with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=eps, max_ipm_iters=i, **kw) as S:
        info = S.solve()
        dm, db = S.scalar("mu"), S.scalar("beta")
        du = S.vector("u").copy()
    same = "" if (o.info["admm_iter"] == info["admm_iter"] and o.work["mu"] == dm) else "   <-- differs"
    print("%3d | %6d  %.17g  %.17g | %6d  %.17g  %.17g  rel(u) %.2e%s" % (i, o.info["admm_iter"], o.work["mu"], o.work["beta"], info["admm_iter"], dm, db,
          np.linalg.norm(np.delete(du, np.s_[A.shape[0]:len(du) - A.shape[1] - 1]) - o.work["u"]) / np.linalg.norm(o.work["u"]) if len(du) >= len(o.work["u"]) else -1, same), flush=True)
