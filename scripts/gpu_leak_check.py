"""GPU: repeated init / solve / finish must not leak device memory (both LP back-ends and the conic path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from abip_amd import Solver, problems, qcp
from test_gpu_qcp import lasso_socp, eps_all
A, b, c = problems.lp_staircase()[:3]
data, K = lasso_socp(200, 600, 3, density=0.05)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
for tag, fn in (("LP direct", lambda: Solver(A, b, c, linsys="direct", verbose=0, max_admm_iters=50).__enter__().__exit__(None, None, None) if False else None),):
    pass
def lp(linsys):
    with Solver(A, b, c, linsys=linsys, verbose=0, max_admm_iters=60) as S:
        S.solve()
from _lasso_cases import gen as lasso_gen
from _svm_cases import gen as svm_gen
from abip_amd import dist as adist
Xl, yl, laml = lasso_gen("wide_sparse"); Xs, ys = svm_gen("tall")
def sharded():
    adist.init_callback(0, 1, lambda arr: None)
    try:
        qcp.abip_qcp(data, K, dict(eps=1e-2, linsys_solver=3, verbose=0))
    finally:
        adist.finalize()
for tag, fn in (("LP direct", lambda: lp("direct")), ("LP indirect", lambda: lp("indirect")), ("QCP", lambda: qcp.abip_qcp(data, K, dict(eps=1e-2, linsys_solver=1, verbose=0))),
                ("QCP PCG", lambda: qcp.abip_qcp(data, K, dict(eps=1e-2, linsys_solver=3, verbose=0))), ("QCP PCG, sharded code path (1 rank)", sharded),
                ("LASSO front end", lambda: qcp.abip_ml(dict(X=Xl, y=yl, **{"lambda": laml}), dict(prob_type=0, eps=1e-2, linsys_solver=1, verbose=0))),
                ("SVM-SOCP front end", lambda: qcp.abip_ml(dict(X=Xs, y=ys, **{"lambda": 1.0}), dict(prob_type=1, eps=1e-2, linsys_solver=1, verbose=0))),
                ("SVM-QP front end, PCG", lambda: qcp.abip_ml(dict(X=Xs, y=ys, **{"lambda": 1e-2}), dict(prob_type=3, eps=1e-2, linsys_solver=3, verbose=0)))):
    for _ in range(5): fn()
    f0 = free()
    for _ in range(150): fn()
    f1 = free()
    print(f"{tag}: free memory change after 150 init/solve/finish cycles: {(f1 - f0) / 1e6:+.2f} MB", flush=True)
