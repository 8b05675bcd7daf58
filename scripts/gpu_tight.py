import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from _golden import *
from abip_amd import Solver
from oracle import pyoracle as po
for name in ("lp_afiro_like", "lp_random_sparse_small", "lp_multicommodity_small"):
    z, A, b, c = load(name)
    for ls in ("indirect", "direct"):
        for eps in (1e-7, 1e-8, 1e-9):
            t=time.time(); o = po.solve("oracle", A, b, c, linsys=ls, eps=eps, max_admm_iters=300000); to=time.time()-t
            with Solver(A, b, c, linsys=ls, verbose=0, eps=eps, max_admm_iters=300000) as S:
                t=time.time(); info = S.solve(); tg=time.time()-t
                print(name, ls, eps, info["status"], o.info["status"], "admm", info["admm_iter"], o.info["admm_iter"], "ipm", info["ipm_iter"], o.info["ipm_iter"],
                      ["%.1e" % rel(getattr(S, k), getattr(o, k)) for k in "xys"], "t %.1f %.1f" % (tg, to), flush=True)
