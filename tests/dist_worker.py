"""Worker for the multi-rank tests (launched by torch.distributed.run; not collected by pytest).

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P tests/dist_worker.py MODE FIXTURE EPS
    python -m torch.distributed.run ...                                                                   tests/dist_worker.py JOBS '[{"mode": .., "fixture": .., "eps": .., "form": ..}, ...]'
(JOBS: several solves in the same processes -- the ranks take ~15 s to start; rank 0 prints the list of results)

MODE = gloo-callback : every rank uses cuda:0 and the host-staged collective over gloo (runs on a 1-GPU box)
MODE = gloo-ordered  : the same with the contributions added in rank order (abip_amd.dist.ordered_sum_allreduce): the peer transport's order
MODE = peer          : every rank uses cuda:0 and the hand-rolled exchange over peer-mapped mailboxes (abip_amd/csrc/dev_peer.h; IPC handles over gloo)
MODE = peer+ordered  : the solve twice in the same processes, over the mailboxes and over the ordered host-staged sums (result of the second under "second");
                       single+peer+ordered: in front of them rank 0 solves the LP on its own, unsharded (under "single")
MODE = rccl          : one GPU per rank, RCCL communicator bootstrapped over torch.distributed
Rank 0 prints a JSON line with the result."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gen_problem(fixture):
    """gen:<kind>:<seed> -- a seeded generator instead of a golden fixture (scripts/gpu_sweep_dist.py; the tests build the same LP to get the single-GPU counts)."""
    from abip_amd import problems
    _, kind, seed = fixture.split(":")
    seed = int(seed)
    rng = np.random.default_rng(seed)
    if kind == "rand":
        m = int(rng.integers(40, 300)); A, b, c = problems.lp_random_sparse(m=m, n=int(m * rng.uniform(1.5, 4)), per_col=int(rng.integers(2, 7)), seed=seed)
    elif kind == "skew":   # a few nearly dense rows on top of a sparse LP: the non-zero-balanced row blocks have very different row counts,
        import scipy.sparse as sp   # so every rank's local grid would differ if it were derived from its own block
        m = int(rng.integers(120, 260)); n = int(m * 3)
        A0, b0, c0 = problems.lp_random_sparse(m=m, n=n, per_col=3, seed=seed)
        dense = sp.random(4, n, density=0.7, random_state=np.random.default_rng(seed + 1), data_rvs=lambda k: np.random.default_rng(seed + 2).uniform(0.1, 1.0, k), format="csr")
        A = sp.vstack([dense, sp.csr_matrix(A0)]).tocsc()
        x0 = np.abs(np.random.default_rng(seed + 3).standard_normal(n)) * (np.random.default_rng(seed + 4).random(n) < 0.4)
        b = A @ x0; c = c0
    elif kind == "odd":    # m, n odd and not multiples of 8 * 32: the chunks of the peer exchange (dev_peer.h peer_chunk: rounded to 32) do not tile the vectors,
        m = 2 * int(rng.integers(150, 400)) + 1   # the last ranks' chunks are short or empty
        A, b, c = problems.lp_random_sparse(m=m, n=2 * int(m * rng.uniform(0.8, 1.6)) + 1, per_col=int(rng.integers(3, 7)), seed=seed)
    elif kind == "stair":
        A, b, c = problems.lp_staircase(seed=seed, stages=int(rng.integers(2, 6)), rows_per=int(rng.integers(8, 30)), cols_per=int(rng.integers(20, 60)))[:3]
    else:
        nd = int(rng.integers(8, 30)); A, b, c = problems.lp_multicommodity(seed=seed, nodes=nd, arcs=int(nd * rng.uniform(2, 4)), commodities=int(rng.integers(2, 5)))[:3]
    return A, b, c


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from _golden import load
    from abip_amd import Solver
    from abip_amd import dist as adist
    if sys.argv[1] == "JOBS":    # several solves in the same processes (ranks take ~15 s to start): a JSON list of {mode, fixture, eps, form}
        jobs = json.loads(sys.argv[2])
    else:
        jobs = [dict(mode=sys.argv[1], fixture=sys.argv[2], eps=float(sys.argv[3]), form=os.environ.get("ABIP_HIP_DIST_CG"))]
    rccl = any(j["mode"] == "rccl" for j in jobs)
    if rccl:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    results = []
    for job in jobs:
        mode, fixture, eps = job["mode"], job["fixture"], float(job["eps"])
        if job.get("form"):
            os.environ["ABIP_HIP_DIST_CG"] = job["form"]      # (read by abip_init)
        else:
            os.environ.pop("ABIP_HIP_DIST_CG", None)
        if fixture.startswith("gen:"):
            A, b, c = gen_problem(fixture)
        else:
            z, A, b, c = load(fixture)

        def transport(which):
            if which == "gloo-callback":
                adist.init_callback(rank, world, lambda arr: dist.all_reduce(torch.from_numpy(arr)))
            elif which == "gloo-ordered":  # the same, the contributions added in rank order: bit-identical to the peer-mapped transport at any world size
                adist.init_callback(rank, world, adist.ordered_sum_allreduce())
            elif which == "peer":          # the hand-rolled exchange over peer-mapped mailboxes (dev_peer.h): every rank on cuda:0, the IPC handles travel over gloo
                adist.init_peer_torch(A.shape[0], A.shape[1])
            else:
                adist.init_torch()

        def run(which):
            transport(which)
            with Solver(A, b, c, linsys="indirect", verbose=0, eps=eps) as S:
                info = S.solve()
                out = dict(rank=rank, world=world, transport=which, fixture=fixture, eps=eps, form=job.get("form"), status=info["status"], admm_iter=info["admm_iter"], ipm_iter=info["ipm_iter"],
                           pobj=info["pobj"], dobj=info["dobj"], cg=S.scalar("tot_cg_its"), cols=S.scalar("dist_cols"), x=S.x.tolist(), y=S.y.tolist(), s=S.s.tolist(), rows=[int(r) for r in S.rows()])
                extra = np.array([S.scalar("mu"), S.scalar("beta"), S.scalar("nb"), S.scalar("tot_cg_its"), float(info["admm_iter"]), info["pobj"]])
            # every rank must hold the same full solution, BIT for bit (replicated n-space state and every host decision derive from all-reduced
            # values and from reductions whose grid is the same on every rank), and the same persistent grid NB
            t = torch.from_numpy(np.concatenate([extra, S.x, S.y, S.s]).astype(np.float64))
            if which == "rccl":
                t = t.cuda()
            gathered = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            out["consistent"] = bool(all(torch.equal(g.view(torch.int64), gathered[0].view(torch.int64)) for g in gathered))
            out["nb"] = float(extra[2])
            rows = [None] * world
            dist.all_gather_object(rows, out["rows"])
            out["rank_rows"] = rows
            adist.finalize()
            return out

        if mode in ("peer+ordered", "single+peer+ordered"):   # both transports in the same processes: the second result rides under "second"
            single = None
            if mode.startswith("single") and rank == 0:   # ... and in front of them the plain one-GPU solve, by rank 0 on its own (the others wait at the barrier)
                with Solver(A, b, c, linsys="indirect", verbose=0, eps=eps) as S1:
                    i1 = S1.solve()
                    single = dict(status=i1["status"], admm_iter=i1["admm_iter"], ipm_iter=i1["ipm_iter"], pobj=i1["pobj"], cg=S1.scalar("tot_cg_its"), x=S1.x.tolist(), y=S1.y.tolist(), s=S1.s.tolist())
            dist.barrier()
            out = run("peer")
            out["second"] = run("gloo-ordered")
            out["single"] = single
            out["shape"] = [int(A.shape[0]), int(A.shape[1])]
        else:
            out = run(mode)
        results.append(out)
    if rank == 0:
        print("RESULT " + json.dumps(results if sys.argv[1] == "JOBS" else results[0]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
