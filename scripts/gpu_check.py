"""Developer diagnostic (GPU box): unit kernels, trajectories and full solves against the CPU oracle, verbose."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from abip_amd import Solver, problems
from oracle import pyoracle as po

def rel(a, b): return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))

def kkt(Asc, rho):
    m, n = Asc.shape
    return sp.bmat([[rho * sp.identity(m), Asc], [Asc.T, -sp.identity(n)]], format="csc")

def unit(name, A, b, c):
    for ls in ("indirect", "direct"):
        with Solver(A, b, c, linsys=ls, verbose=0) as S:
            m, n = S.m, S.n
            Asc = sp.csc_matrix((S.vector("Ax"), A.indices, A.indptr), shape=A.shape)
            rng = np.random.default_rng(0)
            x = rng.standard_normal(n); y = rng.standard_normal(m)
            print(name, ls, "nb", S.scalar("nb"), "A*x", rel(S.accum_by_A(x, y), y + Asc @ x), "At*y", rel(S.accum_by_Atrans(y, x), x + Asc.T @ y))
            rhs = rng.standard_normal(m + n)
            z, its = S.kkt_solve(rhs, None, -1)
            K = kkt(Asc, 1e-3)
            print("   kkt solve its", its, "resid", rel(K @ z, rhs), "lnnz", S.scalar("lnnz"), "lev", S.scalar("levels_fwd"), S.scalar("levels_bwd"), "small", S.scalar("small_solve"))
            z2, its2 = S.kkt_solve(rhs, z[:m] * 1.01, 50)
            print("   kkt warm its", its2, "resid", rel(K @ z2, rhs))

def traj(name, A, b, c, T=12):
    for ls in ("indirect", "direct"):
        o = po.solve("oracle", A, b, c, linsys=ls, eps=1e-9, trace=T, max_admm_iters=100000)
        with Solver(A, b, c, linsys=ls, verbose=0, eps=1e-9) as S:
            S.begin()
            print(name, ls, "g_th", S.scalar("g_th"), o.work["g_th"], "g", rel(S.vector("g"), o.work["g"]))
            for t in range(min(T, len(o.trace))):
                S.step(1)
                print("   it", t + 1, "u", rel(S.vector("u"), o.trace[t, 0]), "v", rel(S.vector("v"), o.trace[t, 1]), "ut", rel(S.vector("u_t"), o.trace[t, 2]),
                      "mu", S.scalar("mu"), "beta", S.scalar("beta"))

def full(name, A, b, c, eps):
    for ls in ("indirect", "direct"):
        t = time.time(); o = po.solve("oracle", A, b, c, linsys=ls, eps=eps); to = time.time() - t
        with Solver(A, b, c, linsys=ls, verbose=0, eps=eps) as S:
            t = time.time(); info = S.solve(); tg = time.time() - t
            print(name, ls, eps, info["status"], "ipm", info["ipm_iter"], o.info["ipm_iter"], "admm", info["admm_iter"], o.info["admm_iter"],
                  "pobj %.10g %.10g" % (info["pobj"], o.info["pobj"]), "x %.2e y %.2e s %.2e" % (rel(S.x, o.x), rel(S.y, o.y), rel(S.s, o.s)),
                  "t_gpu %.2fs t_cpu %.2fs" % (tg, to), "cg", S.scalar("tot_cg_its"), o.work.get("tot_cg_its"))

if __name__ == "__main__":
    P = {"afiro": problems.lp_afiro_like(), "rand300": problems.lp_random_sparse(300, 800, 6, seed=5), "stair": problems.lp_staircase()}
    for k, v in P.items(): unit(k, *v)
    for k in ("afiro", "rand300"): traj(k, *P[k])
    for k, v in P.items():
        for eps in (1e-3, 1e-6): full(k, *v, eps)
