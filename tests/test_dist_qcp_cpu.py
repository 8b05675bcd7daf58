"""CPU, gloo, world_size 2 and 3: the collectives of the column-sharded conic PCG (abip_amd/csrc/qcp_dist.h) on a numpy model -- column blocks from
the product's own partitioner (abip_hip_qcp_dist_partition: cuts behind cones or inside the free / zero / orthant blocks), n-space sharded, m-space
replicated, ONE all-reduce of m doubles per PCG iteration and none for its scalars -- must reproduce the unsharded solve of the conic KKT system,
iteration for iteration, with every rank holding the same y."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pcg_schur(A_cols, Hinv_loc, M, rho_y, by, gx_loc, tol, allreduce, max_its):
    """qcp_pcg.h on a column block: (rho_y I + A H^-1 A') y = by - A H^-1 g_x, then x_g = H_g^-1 (g_x,g + A_g' y).  by, M, y: replicated m-vectors."""
    ncoll = 0
    t = A_cols @ (Hinv_loc * gx_loc); allreduce(t); ncoll += 1                      # kq_prod_A + exchange + kq_dist_prep_fin
    b = by - t
    y = np.zeros_like(b); r = b.copy(); z = M * r; p = z.copy()
    zr = z @ r
    its = 0
    while its < max_its:
        tn = Hinv_loc * (A_cols.T @ p)                                             # kq_pcg_Aty: local
        Gp = A_cols @ tn; allreduce(Gp); ncoll += 1                                 # kq_prod_A + exchange
        Gp = rho_y * p + Gp                                                         # kq_dist_Gp_fin: replicated, as are all the scalars below
        alpha = zr / (p @ Gp)
        y += alpha * p; r -= alpha * Gp
        its += 1
        if np.sqrt(r @ r) < tol:
            break
        z = M * r
        zr_new = z @ r
        p = z + (zr_new / zr) * p
        zr = zr_new
    x_loc = Hinv_loc * (gx_loc + A_cols.T @ y)                                      # kq_pcg_post: local
    return y, x_loc, its, ncoll


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from abip_amd import qcp
    from qcp_cases import make
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data, K = make("mixed")
    A = sp.csc_matrix(data["A"]); m, n = A.shape
    bounds = qcp.partition_columns(A, K, world)
    n0, n1 = int(bounds[rank]), int(bounds[rank + 1])
    rng = np.random.default_rng(3)
    rho_y, rho_x = 1e-3, 1.0
    qd = np.asarray(data["Q"].diagonal())
    Hinv = 1.0 / (rho_x + qd)
    Mpre = 1.0 / (rho_y + np.asarray(A.multiply(A) @ Hinv).ravel())
    by, gx = rng.standard_normal(m), rng.standard_normal(n)
    ar = lambda v: dist.all_reduce(torch.from_numpy(v))
    y, xl, its, ncoll = pcg_schur(A[:, n0:n1], Hinv[n0:n1], Mpre, rho_y, by, gx[n0:n1], 1e-11, ar, m)
    ys = [torch.zeros(m, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(ys, torch.from_numpy(y.copy()))
    xs = [None] * world
    dist.all_gather_object(xs, xl)
    if rank == 0:
        # unsharded reference: the same algorithm with an identity collective, and the KKT system solved directly
        y1, x1, its1, _ = pcg_schur(A, Hinv, Mpre, rho_y, by, gx, 1e-11, lambda v: None, m)
        Kmat = sp.bmat([[rho_y * sp.identity(m), A], [-A.T, sp.diags(1.0 / Hinv)]]).tocsc()      # rho_y y + A x = by ; -A'y + H x = gx
        zz = spla.spsolve(Kmat, np.concatenate([by, gx]))
        q.put(dict(bounds=bounds.tolist(), its=its, its1=its1, ncoll=ncoll, same_y=all(torch.equal(t, ys[0]) for t in ys),
                   dy=float(np.abs(y - y1).max()), dx=float(np.abs(np.concatenate(xs) - x1).max()),
                   dkkt=float(np.abs(np.concatenate([y, np.concatenate(xs)]) - zz).max() / np.abs(zz).max())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_column_sharded_pcg_model(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out["same_y"]                                        # replicated m-space: bit-identical on every rank
    assert out["its"] == out["its1"] and out["ncoll"] == out["its"] + 1      # one collective per PCG iteration (+ the right-hand side), none for scalars
    assert out["dy"] < 1e-10 and out["dx"] < 1e-10 and out["dkkt"] < 1e-8


def test_partition_respects_cones():
    sys.path.insert(0, ROOT)
    from abip_amd import qcp
    from qcp_cases import make
    data, K = make("mixed")
    n = data["A"].shape[1]
    ends, pos = [], 0
    for s in list(K["q"]) + list(K["rq"]):
        pos += s; ends.append(pos)
    for world in (1, 2, 3, 5, 8):
        b = qcp.partition_columns(data["A"], K, world)
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) > 0)
        for cut in b[1:-1]:
            assert cut in ends or cut > pos                    # behind a cone, or inside the free / zero / orthant blocks
    d2, K2 = make("lasso_small")                                # one big cone first: it is never split
    b = qcp.partition_columns(d2["A"], K2, 4)
    assert b[1] >= K2["q"][0]
    with pytest.raises(ValueError):
        qcp.partition_columns(sp.csc_matrix(np.ones((2, 3))), dict(q=[3]), 2)      # a single cone cannot be shared
