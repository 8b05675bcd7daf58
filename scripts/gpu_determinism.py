"""GPU: run-to-run reproducibility.  The design has no atomics and fixed-order reductions, so repeated solves of the same problem in
one process must agree bit for bit, in every mode.  (A cross-workgroup race in kq_ut_prox was found this way.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g
g.build()
from abip_amd import Solver, problems, qcp
from test_gpu_qcp import lasso_socp, toy, eps_all

bad = 0
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lps = {"stair": problems.lp_staircase()[:3], "rand": problems.lp_random_sparse(m=300, n=800, per_col=6, seed=5)[:3], "mc": problems.lp_multicommodity()[:3]}
for name, (A, b, c) in lps.items():
    for linsys in ("direct", "indirect"):
        for batch in ("1", "0"):
            os.environ["ABIP_HIP_BATCH"] = batch
            seen = set()
            for r in range(R):
                with Solver(A, b, c, linsys=linsys, eps=1e-5, verbose=0) as S:
                    info = S.solve()
                    seen.add((info["admm_iter"], S.x.tobytes(), S.y.tobytes()))
            ok = len(seen) == 1
            bad += not ok
            print(f"{'ok ' if ok else 'BAD'} LP {name:6s} {linsys:8s} batch={batch}: {len(seen)} distinct result(s) in {R} runs", flush=True)
for name, (data, K) in {"toy": toy(), "lasso": lasso_socp(400, 1500, 3, density=0.02), "lasso_big": lasso_socp(2200, 2600, 4, density=0.004)}.items():
    for batch in ("1", "0"):
        os.environ["ABIP_HIP_BATCH"] = batch
        seen = set()
        for r in range(R):
            sol, info = qcp.abip_qcp(data, K, eps_all(1e-3 if name == "lasso_big" else 1e-6))
            seen.add((info["admm_iter"], sol["x"].tobytes()))
        ok = len(seen) == 1
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} QCP {name:9s} batch={batch}: {len(seen)} distinct result(s) in {R} runs", flush=True)
print("FAILURES", bad)
