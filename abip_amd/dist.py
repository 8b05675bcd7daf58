"""Multi-GPU set-up for the sharded PCG path (include/abip_hip.h, "multi-GPU"): one process per GPU.

    dist.init_rccl(rank, world, broadcast)   # production: RCCL over xGMI; `broadcast(bytes_or_None) -> bytes` moves the id
    dist.init_torch(process_group=None)      # the same, using torch.distributed for the 128-byte id exchange
    dist.init_callback(rank, world, allreduce)  # tests: host-staged sum through any collective (e.g. gloo)
    dist.init_peer(rank, world, m, n, allgather) # the hand-rolled deterministic exchange over peer-mapped buffers (dev_peer.h); init_peer_torch: over torch.distributed
    dist.finalize()
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib

_keepalive = []


def partition(A, world: int) -> np.ndarray:
    """Row bounds of the sharded path (pure host code; also what abip_init uses)."""
    L = _lib.load()
    A = sp.csc_matrix(A)
    A.sort_indices()
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    Ai = np.ascontiguousarray(A.indices, dtype=np.int64)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
    mat = _lib.ABIPMatrix(Ax.ctypes.data_as(_lib.PF), Ai.ctypes.data_as(_lib.PI), Ap.ctypes.data_as(_lib.PI), A.shape[0], A.shape[1])
    out = np.zeros(world + 1, dtype=np.int64)
    if L.abip_hip_dist_partition(C.byref(mat), world, out.ctypes.data_as(_lib.PI)) != 0:
        raise ValueError("cannot partition: fewer rows than ranks")
    return out


def init_rccl(rank: int, world: int, broadcast) -> None:
    L = _lib.load()
    buf = C.create_string_buffer(128)
    if rank == 0 and L.abip_hip_dist_get_unique_id(buf) != 0:
        raise RuntimeError("ncclGetUniqueId failed (librccl not loadable?)")
    uid = broadcast(buf.raw if rank == 0 else None)
    idbuf = C.create_string_buffer(bytes(uid), 128)
    rc = L.abip_hip_dist_init_rccl(rank, world, idbuf)
    if rc != 0:
        raise RuntimeError(f"abip_hip_dist_init_rccl failed ({rc})")


def init_torch(process_group=None) -> None:
    """RCCL communicator for the solver, bootstrapped over an existing torch.distributed process group."""
    import torch.distributed as dist
    rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)

    def bcast(payload):
        box = [payload]
        dist.broadcast_object_list(box, src=0, group=process_group)
        return box[0]

    init_rccl(rank, world, bcast)


def init_callback(rank: int, world: int, allreduce) -> None:
    """`allreduce(np.ndarray)` must sum the array in place over all ranks (host memory)."""
    L = _lib.load()

    def _cb(_ctx, ptr, count):
        arr = np.ctypeslib.as_array(ptr, shape=(count,))
        allreduce(arr)

    fn = _lib.ALLREDUCE_FN(_cb)
    _keepalive.append(fn)
    if L.abip_hip_dist_init_callback(rank, world, fn, None) != 0:
        raise RuntimeError("abip_hip_dist_init_callback failed")


def init_peer(rank: int, world: int, m: int, n: int, allgather) -> None:
    """The hand-rolled exchange over peer-mapped mailboxes (abip_amd/csrc/dev_peer.h).  `allgather(bytes) -> list[bytes]` (in rank order) moves the
    64-byte IPC handles, e.g. over torch.distributed.all_gather_object; (m, n) size the mailbox for the LP about to be solved."""
    L = _lib.load()
    cap = int(L.abip_hip_dist_peer_capacity(int(m), int(n)))
    buf = C.create_string_buffer(64)
    rc = L.abip_hip_dist_peer_prepare(cap, buf)
    if rc != 0:
        raise RuntimeError(f"abip_hip_dist_peer_prepare failed ({rc})")
    handles = allgather(buf.raw)
    if len(handles) != world or any(len(h) != 64 for h in handles):
        raise RuntimeError("the handle exchange did not return one 64-byte handle per rank")
    allh = C.create_string_buffer(b"".join(handles), 64 * world)
    rc = L.abip_hip_dist_init_peer(rank, world, allh)
    # every rank must come to the same conclusion: one that refuses (-4: a coarse-grained mailbox with a peer on another device, no coherence guarantee) while
    # the others go on would leave them waiting in the first exchange
    rcs = [int.from_bytes(b[:4], "little", signed=True) for b in allgather(int(rc).to_bytes(4, "little", signed=True) + bytes(60))]
    if any(r != 0 for r in rcs):
        L.abip_hip_dist_finalize()
        raise RuntimeError(f"abip_hip_dist_init_peer failed on rank(s) {[q for q, r in enumerate(rcs) if r != 0]} (codes {rcs}); "
                           "-4 = the mailbox is not fine-grained and a peer sits on another device: use the RCCL transport (dist.init_torch)")


def init_peer_torch(m: int, n: int, process_group=None) -> None:
    import torch.distributed as dist
    rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)

    def gather(payload):
        box = [None] * world
        dist.all_gather_object(box, payload, group=process_group)
        return box

    init_peer(rank, world, m, n, gather)


def ordered_sum_allreduce(process_group=None):
    """A host-staged all-reduce for dist.init_callback that adds the ranks' contributions IN RANK ORDER ((r0 + r1) + r2 + ...): the order the peer-mapped
    transport uses (dev_peer.h), so a run over this callback is bit-identical to a run over the mailboxes at any world size (gloo's own ring all-reduce
    associates three or more terms in another order)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(process_group)

    def allreduce(arr: np.ndarray) -> None:
        mine = torch.from_numpy(arr)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=process_group)
        acc = parts[0].clone()
        for q in range(1, world):
            acc += parts[q]
        arr[:] = acc.numpy()

    return allreduce


def comm_count() -> int:
    """Ranks of the live communicator as the transport itself reports them (RCCL: ncclCommCount); 0 without one."""
    return int(_lib.load().abip_hip_dist_comm_count())


def finalize() -> None:
    _lib.load().abip_hip_dist_finalize()
    _keepalive.clear()
