import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abip_amd import problems, qcp
p, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 45000)
data, K = problems.qcp_lasso_socp(p, d)
sol, info = qcp.abip_qcp(data, K, dict(eps=float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3, linsys_solver=1, verbose=0))
print(info)
