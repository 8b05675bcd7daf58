"""Developer check: the one-XCD launch against the launch path, iteration by iteration, on a golden fixture."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _golden import load
from abip_amd.solver import Solver

name = sys.argv[1] if len(sys.argv) > 1 else "lp_tiny_scale5"
linsys = sys.argv[2] if len(sys.argv) > 2 else "indirect"
nit = int(sys.argv[3]) if len(sys.argv) > 3 else 60
kw = {}
if name.endswith("scale5"): kw = dict(scale=5.0)
z, A, b, c = load(name)
sol = {}
for xcd in (1, 0):
    os.environ["ABIP_HIP_XCD"] = str(xcd)
    S = Solver(A, b, c, linsys=linsys, verbose=0, eps=1e-4, **kw)
    S.begin()
    tr = []
    for it in range(nit):
        fin, done = S.step(1)
        tr.append((S.vector("u").copy(), S.vector("v").copy(), S.scalar("tot_cg_its"), S.scalar("admm_iter"), S.scalar("mu")))
        if fin: break
    sol[xcd] = tr
    print("xcd", xcd, "on", S.scalar("xcd"), "steps", len(tr))
    S.close()
for i, (a, b_) in enumerate(zip(sol[1], sol[0])):
    du = np.linalg.norm(a[0] - b_[0]) / max(1e-300, np.linalg.norm(b_[0]))
    dv = np.linalg.norm(a[1] - b_[1]) / max(1e-300, np.linalg.norm(b_[1]))
    print(i, "du %.2e dv %.2e cg %d/%d k %d/%d mu %.3e/%.3e" % (du, dv, a[2], b_[2], a[3], b_[3], a[4], b_[4]))
