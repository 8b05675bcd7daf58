"""CPU, gloo, world_size 2: the collectives of the row-sharded PCG path on a numpy model --
row blocks from the product's own partitioner (abip_hip_dist_partition), local A_g'y_g partials all-reduced, m-space
dots all-reduced, n-space replicated -- must reproduce the unsharded solve of K z = rhs, iteration for iteration."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pcg_kkt(A_rows, rhs_y, rhs_x, rho, tol, allreduce, m_glob):
    """indirect.c:393-434 / 321-391 on a row block, with the communication pattern of the device path (solver.hip:
    enqueue_cg_chunk): ONE collective per CG iteration carrying [A_g'z_g | r'r, z'r, z'z, z'p]; p'Gp is rebuilt as
    rho ||p||^2 + ||A'p||^2 from the replicated A'p and the recurrence ||z + beta p||^2 = z'z + 2 beta z'p + beta^2 ||p||^2.
    Returns (y_block, x, iterations, collectives)."""
    n = A_rows.shape[1]
    ncoll = 0
    M = 1.0 / np.asarray(A_rows.multiply(A_rows).sum(axis=1)).ravel()
    b = rhs_y + A_rows @ rhs_x
    x = np.zeros_like(b); r = b.copy(); z = M * r; p = np.zeros_like(b)
    tmp = np.zeros(n); pp = 0.0; zr_old = 1.0
    its = 0
    for its in range(1, m_glob + 1):
        pack = np.concatenate([A_rows.T @ z, [r @ r, z @ r, z @ z, z @ p]])
        allreduce(pack); ncoll += 1
        rr, zr, zz, zp = pack[n:]
        if its > 1 and np.sqrt(rr) < tol:   # the stopping test of the previous update, seen one collective later
            its -= 1
            break
        beta = 0.0 if its == 1 else zr / zr_old
        tmp = pack[:n] + beta * tmp          # A'(z + beta p), replicated
        pp = zz + 2.0 * beta * zp + beta * beta * pp
        p = z + beta * p
        Gp = A_rows @ tmp + rho * p
        alpha = zr / (rho * pp + tmp @ tmp)
        x += alpha * p; r -= alpha * Gp
        z = M * r
        zr_old = zr
    t = A_rows.T @ x
    allreduce(t); ncoll += 1
    return x, t - rhs_x, its, ncoll


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from abip_amd import dist as adist, problems
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b, c = problems.lp_random_sparse(m=180, n=420, per_col=5, seed=21)
    A = sp.csr_matrix(A)
    bounds = adist.partition(A, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(A.shape[0] + A.shape[1])
    ar = lambda arr: dist.all_reduce(torch.from_numpy(arr))
    y_blk, x, its, ncoll = pcg_kkt(A[r0:r1], rhs[r0:r1], rhs[A.shape[0]:], 1e-3, 1e-10, ar, A.shape[0])
    y_full = np.zeros(A.shape[0]); y_full[r0:r1] = y_blk; ar(y_full)
    if rank == 0:
        q.put((bounds.tolist(), y_full, x, its, ncoll))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_pcg_equals_unsharded(world):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from abip_amd import problems
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29871 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    bounds, y, x, its, ncoll = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A, b, c = problems.lp_random_sparse(m=180, n=420, per_col=5, seed=21)
    A = sp.csr_matrix(A)
    m, n = A.shape
    assert bounds[0] == 0 and bounds[-1] == m and all(b2 > b1 for b1, b2 in zip(bounds, bounds[1:]))
    nnz_blk = [A[bounds[g]:bounds[g + 1]].nnz for g in range(world)]
    assert max(nnz_blk) <= 1.25 * (A.nnz / world) + 16          # balanced by non-zeros
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(m + n)
    K = sp.bmat([[1e-3 * sp.identity(m), A], [A.T, -sp.identity(n)]], format="csc")
    ref = spla.spsolve(K, rhs)
    assert np.linalg.norm(np.concatenate([y, x]) - ref) / np.linalg.norm(ref) < 1e-7
    y1, x1, its1, _ = pcg_kkt(A, rhs[:m], rhs[m:], 1e-3, 1e-10, lambda arr: None, m)     # one "rank", no communication
    assert its == its1 and ncoll == its + 2          # one collective per CG iteration (+ the late stopping test, + the back-substitution)
    assert np.linalg.norm(y - y1) / np.linalg.norm(y1) < 1e-10 and np.linalg.norm(x - x1) / np.linalg.norm(x1) < 1e-10


def test_partition_edge_cases():
    sys.path.insert(0, ROOT)
    from abip_amd import dist as adist
    A = sp.identity(5, format="csc")
    assert adist.partition(A, 5).tolist() == [0, 1, 2, 3, 4, 5]
    assert adist.partition(A, 1).tolist() == [0, 5]
    with pytest.raises(ValueError):
        adist.partition(A, 6)
    # one dense row must not starve the other ranks of rows
    B = sp.vstack([sp.csr_matrix(np.ones((1, 50))), sp.random(9, 50, density=0.05, random_state=1)]).tocsc()
    bd = adist.partition(B, 4).tolist()
    assert bd[0] == 0 and bd[-1] == 10 and all(b2 > b1 for b1, b2 in zip(bd, bd[1:]))


# ---- the COLUMN form of the sharded solve (ABIP_HIP_DIST_CG=cols; solver.hip enqueue_cg_* with w->cg_cols) on the same numpy model --------------
def pcg_kkt_cols(A_rows, A_cols, row0, rhs_y_blk, rhs_x, c0, rho, tol, allreduce, m_glob):
    """The iteration around the solve keeps the row blocks (rhs_y arrives as the rank's rows, x is replicated); inside the solve the m-space is
    gathered and replicated and A is used by its column block A_cols = A[:, c0:c1]: A'p is local, A (A'p) is ONE exchange of m doubles, the PCG's
    scalars need none.  Collectives: gather of b_y, A b_x, one per iteration, the back-substitution (an n-vector, as in the row form)."""
    n = rhs_x.size
    ncoll = 0
    by = np.zeros(m_glob); by[row0:row0 + rhs_y_blk.size] = rhs_y_blk
    allreduce(by); ncoll += 1                                         # k_cols_place + exchange
    M = np.zeros(m_glob); M[:] = np.asarray(A_cols.multiply(A_cols).sum(axis=1)).ravel()
    allreduce(M); M = 1.0 / M                                          # (set-up, once: the product computes it on the host from the whole matrix)
    nc = A_cols.shape[1]
    t = A_cols @ rhs_x[c0:c0 + nc]; allreduce(t); ncoll += 1           # k_spmv_set(dAc, b_x block) + exchange; k_cols_init_fin
    b = by + t
    y = np.zeros(m_glob); r = b.copy(); z = M * r; p = np.zeros(m_glob); tmp = np.zeros(nc); zr_old = 1.0
    its = 0
    for its in range(1, m_glob + 1):
        rr, zr = r @ r, z @ r                                          # replicated: no exchange
        if its > 1 and np.sqrt(rr) < tol:
            its -= 1
            break
        beta = 0.0 if its == 1 else zr / zr_old
        tmp = A_cols.T @ z + beta * tmp                                # k_cg_spmv_At on the column block: local
        Gp = A_cols @ tmp; allreduce(Gp); ncoll += 1                   # k_spmv_set(dAc) + exchange
        p = z + beta * p; Gp = Gp + rho * p                            # k_cols_Gp_fin
        alpha = zr / (p @ Gp)
        y += alpha * p; r -= alpha * Gp; z = M * r                     # k_cg_update
        zr_old = zr
    T = np.zeros(n); T[c0:c0 + nc] = A_cols.T @ y
    allreduce(T); ncoll += 1                                           # k_cols_place into T + exchange; k_dist_post
    return y[row0:row0 + rhs_y_blk.size], T - rhs_x, its, ncoll


def _worker_cols(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from abip_amd import dist as adist, problems
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b, c = problems.lp_random_sparse(m=180, n=420, per_col=5, seed=21)
    A = sp.csr_matrix(A); Ac = sp.csc_matrix(A)
    m, n = A.shape
    bounds = adist.partition(A, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    cb = np.linspace(0, n, world + 1).astype(int)                      # (the product balances the column blocks by non-zeros; any contiguous cut works)
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(m + n)
    ar = lambda arr: dist.all_reduce(torch.from_numpy(arr))
    y_blk, x, its, ncoll = pcg_kkt_cols(A[r0:r1], Ac[:, cb[rank]:cb[rank + 1]], r0, rhs[r0:r1], rhs[m:], int(cb[rank]), 1e-3, 1e-10, ar, m)
    y_full = np.zeros(m); y_full[r0:r1] = y_blk; ar(y_full)
    xs = [None] * world
    dist.all_gather_object(xs, x)
    if rank == 0:
        q.put((y_full, x, its, ncoll, all(np.array_equal(xs[0], t) for t in xs)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_column_form_of_the_sharded_pcg_equals_unsharded(world):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from abip_amd import problems
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cols, args=(r, world, 29880 + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    y, x, its, ncoll, same_x = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A, b, c = problems.lp_random_sparse(m=180, n=420, per_col=5, seed=21)
    A = sp.csr_matrix(A)
    m, n = A.shape
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(m + n)
    K = sp.bmat([[1e-3 * sp.identity(m), A], [A.T, -sp.identity(n)]], format="csc")
    ref = spla.spsolve(K, rhs)
    assert same_x and np.linalg.norm(np.concatenate([y, x]) - ref) / np.linalg.norm(ref) < 1e-7
    y1, x1, its1, _ = pcg_kkt(A, rhs[:m], rhs[m:], 1e-3, 1e-10, lambda arr: None, m)     # the row form's model on one "rank"
    assert its == its1 and ncoll == its + 3          # one exchange of m doubles per PCG iteration + gather of b_y, A b_x, the back-substitution
    assert np.linalg.norm(y - y1) / np.linalg.norm(y1) < 1e-9 and np.linalg.norm(x - x1) / np.linalg.norm(x1) < 1e-9


# ---- the rank-ordered host-staged sum (abip_amd.dist.ordered_sum_allreduce): the order of the peer-mapped transport (dev_peer.h), any world size ----
def _worker_ordered(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from abip_amd import dist as adist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar = adist.ordered_sum_allreduce()
    v = np.random.default_rng(100 + rank).standard_normal(1001) * 10.0 ** np.random.default_rng(200 + rank).integers(-8, 8, 1001)
    ar(v)
    q.put((rank, v))
    dist.destroy_process_group()


def test_ordered_sum_allreduce_adds_in_rank_order():
    """((r0 + r1) + r2) + ... on every rank, bit for bit: what lets the GPU tests hold the peer-mapped transport to the host-staged one with 4 and 8 ranks."""
    import torch.multiprocessing as mp
    world = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ordered, args=(r, world, 29899, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = [np.random.default_rng(100 + r).standard_normal(1001) * 10.0 ** np.random.default_rng(200 + r).integers(-8, 8, 1001) for r in range(world)]
    want = parts[0].copy()
    for r in range(1, world):
        want += parts[r]
    for r in range(world):
        assert np.array_equal(got[r], want), r
