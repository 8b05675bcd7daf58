import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _golden import *
from abip_amd import Solver
for v in sorted(TINY_VARIANTS):
    z, A, b, c = load("lp_tiny_" + v)
    for ls in ("indirect", "direct"):
        tag = f"{ls}_0.0001"; g = info_of(z, tag)
        with Solver(A, b, c, linsys=ls, verbose=0, eps=1e-4, **TINY_VARIANTS[v]) as S:
            info = S.solve()
            print(v, ls, "admm", info["admm_iter"], int(g["admm_iter"]), "ipm", info["ipm_iter"], int(g["ipm_iter"]), ["%.2e" % rel(getattr(S, k), z[f"{tag}_{k}"]) for k in "xys"])
