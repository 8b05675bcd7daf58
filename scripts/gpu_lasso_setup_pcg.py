import os, time
os.environ["ABIP_HIP_SETUP_TIMES"] = "1"
from abip_amd import problems, qcp
X, y, lam = problems.lasso_protocol_data(5000, 15000)
for rep in range(2):
    sol, info = qcp.abip_ml(dict(X=X, y=y, **{"lambda": lam}), dict(prob_type=0, eps=1e-3, linsys_solver=3, verbose=0))
    print("setup %.3f solve %.3f" % (info["setup_time"], info["solve_time"]))
