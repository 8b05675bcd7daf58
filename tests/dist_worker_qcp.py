"""Worker for the multi-rank tests of the CONIC path (launched by torch.distributed.run; not collected by pytest).

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P tests/dist_worker_qcp.py MODE CASE EPS

MODE = gloo-callback : every rank uses cuda:0 and the host-staged collective over gloo (runs on a 1-GPU box)
MODE = rccl          : one GPU per rank, RCCL communicator bootstrapped over torch.distributed
Every rank hands abip_qcp() the WHOLE problem (as every rank of the LP path does); the library cuts out the rank's column block.
Rank 0 prints a JSON line with the result; `consistent` says that every rank returned the same (x, y, s, info), bit for bit."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def case(name):
    from qcp_cases import make
    return make(name)


def main():
    mode, name, eps = sys.argv[1], sys.argv[2], float(sys.argv[3])
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from abip_amd import dist as adist
    from abip_amd import qcp
    ml = name.startswith("ml:")          # ml:<prob_type>:<case of tests/_lasso_cases.py or tests/_svm_cases.py>
    if ml:
        _, pt, cname = name.split(":")
        pt = int(pt)
        if pt == 0:
            from _lasso_cases import gen
            X, yv, lam = gen(cname)
        else:
            from _svm_cases import gen
            X, yv = gen(cname)
            lam = 1e-2
    else:
        data, K = case(name)
    if mode == "gloo-callback":
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        adist.init_callback(rank, world, lambda arr: dist.all_reduce(torch.from_numpy(arr)))
    else:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", rank=rank, world_size=world)
        adist.init_torch()
    if ml:
        sol, info = qcp.abip_ml(dict(X=X, y=yv, **{"lambda": lam}), dict(prob_type=pt, eps=eps, linsys_solver=3, verbose=0))
        sol = dict(x=sol["x"], y=np.atleast_1d(sol.get("b", 0.0)), s=sol.get("xi", np.zeros(1)))
    else:
        sol, info = qcp.abip_qcp(data, K, dict(eps=eps, linsys_solver=3, verbose=0))
    out = dict(rank=rank, world=world, status=info["status"], admm_iter=info["admm_iter"], ipm_iter=info["ipm_iter"], pobj=info["pobj"], dobj=info["dobj"],
               avg_cg_iters=info["avg_cg_iters"], collectives=info["factor"]["head_nnz"], x=sol["x"].tolist(), y=sol["y"].tolist(), s=sol["s"].tolist())
    t = torch.from_numpy(np.concatenate([[float(info["admm_iter"]), info["pobj"], info["dobj"], info["res_pri"], info["res_dual"], info["gap"]], sol["x"], sol["y"], sol["s"]]).astype(np.float64))
    if mode != "gloo-callback":
        t = t.cuda()
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    out["consistent"] = bool(all(torch.equal(g.view(torch.int64), gathered[0].view(torch.int64)) for g in gathered))
    if rank == 0:
        print("RESULT " + json.dumps(out), flush=True)
    adist.finalize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
