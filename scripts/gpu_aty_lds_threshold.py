"""GPU: where the LDS-resident A'z kernel (ABIP_HIP_ATY_LDS) starts to pay -- the conic PCG back-end on LASSO-as-SOCP at growing sizes, streaming kernel vs LDS kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abip_amd import problems, qcp

for (p, d) in [(1000, 4500), (2000, 9000), (3500, 15750), (5000, 22500)]:
    data, K = problems.qcp_lasso_socp(p, d)
    row = []
    for mode in ("0", "1"):
        os.environ["ABIP_HIP_ATY_LDS"] = mode
        best = None
        for rep in range(2):
            sol, info = qcp.abip_qcp(data, K, dict(eps=1e-3, linsys_solver=3, verbose=0))
            r = info["admm_iter"] / max(info["solve_time"], 1e-9)
            best = r if best is None else max(best, r)
        row.append((best, info["admm_iter"], info["avg_cg_iters"]))
    print(f"p={p} d={d} nnz={data['A'].nnz}: streaming {row[0][0]:.0f} it/s, LDS {row[1][0]:.0f} it/s (admm {row[0][1]}/{row[1][1]}, cg {row[0][2]:.1f})", flush=True)
