import time, numpy as np
from abip_amd import problems, qcp
X,y,lam=problems.lasso_data()
for ls in (1,3):
    for rep in range(2):
        t=time.time(); sol,info=qcp.abip_ml(dict(X=X,y=y,**{"lambda":lam}),dict(prob_type=0,eps=1e-3,linsys_solver=ls,verbose=0)); t=time.time()-t
    print("LASSO special ls",ls,info["status"],info["ipm_iter"],info["admm_iter"],"setup %.3f solve %.3f wall %.3f"%(info["setup_time"],info["solve_time"],t),"pobj",info["pobj"],"cg",info["avg_cg_iters"],"nnz beta",int((np.abs(sol["x"])>1e-6).sum()), info["factor"])
data,K=problems.qcp_lasso_socp()
for ls in (1,3):
    sol,info=qcp.abip_qcp(data,K,dict(eps=1e-3,linsys_solver=ls,verbose=0))
    print("generic ls",ls,info["status"],info["ipm_iter"],info["admm_iter"],"setup %.3f solve %.3f"%(info["setup_time"],info["solve_time"]),"pobj",info["pobj"])
