"""GPU (-m gpu): the row-sharded PCG path.

Only ONE GPU is available to these tests, so the N>1 sharding is exercised with the host-staged collective
(abip_hip_dist_init_callback) over gloo with both ranks on cuda:0 -- the same kernels, partials, gather and
scalar exchange as the RCCL path, only the transport differs -- and the RCCL transport itself with a 1-rank
communicator.  The driver's 2/4/8-GPU bench runs the RCCL transport for real."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from _golden import info_of, load, rel

pytestmark = pytest.mark.gpu
def _free_port():
    """A port nobody holds (hash() of a tuple with a str is salted per interpreter: neither reproducible nor collision-free)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
    import abip_amd
    return abip_amd


def _single(gpu, A, b, c, eps):
    with gpu.Solver(A, b, c, linsys="indirect", verbose=0, eps=eps) as S:
        info = S.solve()
        return info, S.x.copy(), S.y.copy(), S.s.copy(), S.scalar("tot_cg_its")


@pytest.mark.parametrize("name,eps", [("lp_afiro_like", 1e-6), ("lp_random_sparse_small", 1e-6), ("lp_multicommodity_small", 1e-4)])
def test_one_rank_sharded_path_equals_plain_path(gpu, name, eps):
    """world = 1 through the sharded code path (fold kernels, T buffer, split SpMV + combine kernels) with an identity collective."""
    from abip_amd import dist as adist
    z, A, b, c = load(name)
    ref = _single(gpu, A, b, c, eps)
    adist.init_callback(0, 1, lambda arr: None)
    try:
        got = _single(gpu, A, b, c, eps)
    finally:
        adist.finalize()
    assert got[0]["status_val"] == ref[0]["status_val"] == 1
    assert got[0]["admm_iter"] == ref[0]["admm_iter"] and got[0]["ipm_iter"] == ref[0]["ipm_iter"] and got[4] == ref[4]
    for a, r in zip(got[1:4], ref[1:4]):
        assert rel(a, r) < 1e-9


def test_one_rank_rccl_transport(gpu):
    from abip_amd import dist as adist
    z, A, b, c = load("lp_random_sparse_small")
    ref = _single(gpu, A, b, c, 1e-6)
    adist.init_rccl(0, 1, lambda payload: payload)
    try:
        got = _single(gpu, A, b, c, 1e-6)
    finally:
        adist.finalize()
    assert got[0]["admm_iter"] == ref[0]["admm_iter"] and got[4] == ref[4]
    for a, r in zip(got[1:4], ref[1:4]):
        assert rel(a, r) < 1e-9


@pytest.mark.parametrize("form", ["rows", "cols"])
def test_one_rank_both_forms_of_the_sharded_pcg(gpu, monkeypatch, form):
    """The sharded solve has two forms: rows (default, north_star's: A by row blocks, one exchange of n doubles + packed scalars per PCG iteration) and
    columns (ABIP_HIP_DIST_CG=cols: inside the solve the m-space is gathered and replicated and A is used by column blocks, one exchange of m doubles per
    PCG iteration).  Every other test of this file runs the default; this one and the next name the form.  world = 1 with an identity collective against
    the plain path."""
    from abip_amd import dist as adist
    z, A, b, c = load("lp_random_sparse_small")
    ref = _single(gpu, A, b, c, 1e-6)
    monkeypatch.setenv("ABIP_HIP_DIST_CG", form)
    adist.init_callback(0, 1, lambda arr: None)
    try:
        got = _single(gpu, A, b, c, 1e-6)
    finally:
        adist.finalize()
    assert got[0]["status_val"] == ref[0]["status_val"] == 1
    assert got[0]["ipm_iter"] == ref[0]["ipm_iter"] and got[0]["admm_iter"] == ref[0]["admm_iter"], (got[0]["admm_iter"], ref[0]["admm_iter"])
    for a, r in zip(got[1:4], ref[1:4]):
        assert rel(a, r) < 1e-5


# ---- two and three ranks on the one GPU: ONE launch per world size runs every case below in the same processes (tests/dist_worker.py JOBS -- the ranks take
# ~10 s to start, the solves a few seconds each); the tests pick their results out of it ----
JOBS = {
    2: [dict(mode="gloo-callback", fixture="lp_random_sparse_small", eps=1e-6, form="rows"), dict(mode="gloo-callback", fixture="lp_random_sparse_small", eps=1e-3, form="cols"),
        dict(mode="peer+ordered", fixture="lp_random_sparse_small", eps=1e-3, form="rows"), dict(mode="peer+ordered", fixture="lp_afiro_like", eps=1e-6, form="cols"),
        dict(mode="gloo-callback", fixture="gen:skew:11", eps=1e-4, form=None)],
    3: [dict(mode="gloo-callback", fixture="lp_afiro_like", eps=1e-6, form="rows"), dict(mode="gloo-callback", fixture="lp_afiro_like", eps=1e-6, form="cols"),
        dict(mode="peer+ordered", fixture="lp_afiro_like", eps=1e-6, form="rows"), dict(mode="peer+ordered", fixture="gen:skew:11", eps=1e-4, form="cols"),
        dict(mode="gloo-callback", fixture="gen:skew:11", eps=1e-4, form=None)],
}


@pytest.fixture(scope="module")
def runs(gpu):
    cache = {}

    def get(world):
        if world not in cache:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "JOBS", json.dumps(JOBS[world])]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ABIP_HIP_PEER_WAIT_MS="60000")   # (the ranks share one device: a peer may be off it for a while)
            env.pop("ABIP_HIP_DIST_CG", None)
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            assert p.returncode == 0 and lines, p.stdout[-2000:] + p.stderr[-3000:]
            cache[world] = json.loads(lines[-1][7:])
        return cache[world]
    return get


def _against_the_fixture(out, name, eps):
    z, A, b, c = load(name)
    g = info_of(z, f"indirect_{eps:g}")
    assert out["consistent"] and out["status"] == "Solved"
    assert out["ipm_iter"] == g["ipm_iter"] and out["admm_iter"] == g["admm_iter"], (out["admm_iter"], g["admm_iter"])
    for k in "xys":
        assert rel(np.array(out[k]), z[f"indirect_{eps:g}_{k}"]) < 10 * eps, k
    assert abs(out["pobj"] - g["pobj"]) <= 10 * eps * (1 + abs(g["pobj"]))


@pytest.mark.parametrize("world,job", [(2, 0), (2, 1), (3, 0), (3, 1)])
def test_multi_rank_both_forms_match_reference(runs, world, job):
    """Rows and columns form of the sharded PCG with 2 and 3 ranks (host-staged sums over gloo): every rank the same bits, the reference's counts, (x, y, s) within
    10 eps of its fixture (eps 1e-6 with 2 and with 3 ranks; the column form with 2 ranks at 1e-3: the host-staged sums make a long solve slow)."""
    spec = JOBS[world][job]
    out = runs(world)[job]
    assert out["fixture"] == spec["fixture"] and (spec["form"] is None or out["cols"] == (1.0 if spec["form"] == "cols" else 0.0))
    _against_the_fixture(out, spec["fixture"], spec["eps"])


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_skewed_row_blocks_stay_bit_identical(runs, world):
    """Row blocks balanced by non-zeros can hold very different numbers of rows (a few nearly dense rows on one rank).  The replicated
    n-space reductions must still add in the same order on every rank: the persistent grid is derived from global quantities only, so
    x, y, s, mu, beta, the CG count -- everything -- is bit-identical across the ranks ('consistent'), and NB is the same number."""
    out = runs(world)[4]
    assert out["fixture"] == "gen:skew:11" and out["consistent"] and out["status"] == "Solved" and out["nb"] >= 1
    rows = out["rank_rows"]
    assert max(r[1] - r[0] for r in rows) > 2 * min(r[1] - r[0] for r in rows)     # (the blocks ARE skewed)


@pytest.mark.parametrize("world,job", [(2, 2), (2, 3), (3, 2), (3, 3)])
def test_peer_mapped_exchange_is_a_drop_in_for_the_collective(runs, world, job):
    """The hand-rolled transport (abip_amd/csrc/dev_peer.h: one-shot reduce-scatter + all-gather over IPC-mapped mailboxes, every chunk summed in one place
    in rank order) under the same sharded solve: 2 or 3 processes on the one GPU, the handles exchanged over gloo.  Every rank holds the same bits
    ('consistent'), and the run is BIT-identical to the host-staged transport that adds the contributions in rank order (abip_amd.dist.ordered_sum_allreduce),
    with 2 ranks and with 3 (round 5 compared with gloo's own ring sum, which associates three terms differently: counts only)."""
    spec = JOBS[world][job]
    a = runs(world)[job]
    b = a["second"]
    assert a["fixture"] == spec["fixture"] and a["transport"] == "peer" and b["transport"] == "gloo-ordered" and a["cols"] == (1.0 if spec["form"] == "cols" else 0.0)
    assert a["consistent"] and b["consistent"] and a["status"] == b["status"] == "Solved"
    assert a["ipm_iter"] == b["ipm_iter"] and a["admm_iter"] == b["admm_iter"] and a["cg"] == b["cg"]
    for k in "xys":
        assert np.array_equal(np.array(a[k]), np.array(b[k])), k
    if not spec["fixture"].startswith("gen:"):
        _against_the_fixture(a, spec["fixture"], spec["eps"])


def test_coarse_mailbox_with_peers_on_other_devices_is_refused(gpu):
    """when the mailbox cannot be fine-grained memory (here: ABIP_HIP_PEER_FINE=0) and a peer sits on another device (pretended through the hooks
    library on this one-GPU box), remote writes into memory a local kernel polls have no coherence guarantee: abip_hip_dist_init_peer refuses (-4), every rank
    hears of it, nobody starts a solve that would wait for ever."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), "peer", "lp_afiro_like", "1e-3"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ABIP_HIP_PEER_FINE="0", ABIP_HIP_PEER_PRETEND_CROSS="1",
               ABIP_HIP_LIBRARY=os.path.join(ROOT, "abip_amd", "lib", "libabip_hip_hooks.so"))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert "the peer-mapped transport is refused" in p.stderr and "use the RCCL transport" in p.stderr
    # ... and the same two ranks with the override go through (the ranks do share the device: nothing is wrong with the memory)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(env, ABIP_HIP_PEER_COARSE_OK="1"), cwd=ROOT)
    assert p.returncode == 0 and [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")], p.stderr[-2000:]
