"""GPU (-m gpu): the sharded solve with FOUR and EIGHT ranks on the one GPU of the test box (the metric is quoted at 1 / 2 / 4 / 8 GPUs and nothing
above three ranks had ever run), through tests/dist_worker.py and through bench.py --gpus 8 itself.

This file sorts in front of the other GPU tests ON PURPOSE and never touches the GPU in the pytest process: eight worker processes whose kernels wait for each other
(the peer-mapped mailboxes: dev_peer.h) need all eight resident on the device at once, and the driver schedules eight processes side by side -- with a ninth that
holds a context (what the pytest process is once any in-process GPU test has run) the same run takes 3 to 40 times as long (profiles/r06_eight_rank_dry_run.txt).
On a real node every rank has a GPU of its own and none of this arises."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from _golden import info_of, load, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


@pytest.fixture(scope="module")
def built():
    """The libraries, built (or found built) by a CHILD: __graft_entry__.build() loads them, which this process must not do before the eight-rank runs."""
    p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return True


def _run_jobs(world, jobs, timeout=900):
    """One launch of `world` ranks running every job of the list in the same processes (tests/dist_worker.py JOBS): the ranks take ~15 s to start."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "JOBS", json.dumps(jobs)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ABIP_HIP_PEER_WAIT_MS="60000")   # (eight ranks take turns on one device: a peer may be off it for a while)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert p.returncode == 0 and lines, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads(lines[-1][7:])


CASES = {8: [("lp_random_sparse_small", 1e-3, "rows"), ("gen:odd:7", 1e-3, "cols")], 4: [("gen:odd:7", 1e-3, "rows"), ("lp_random_sparse_small", 1e-3, "cols")]}


@pytest.fixture(scope="module")
def runs(built):
    """world -> results of its cases, one launch per world size (run on first use)."""
    cache = {}

    def get(world):
        if world not in cache:
            cache[world] = _run_jobs(world, [dict(mode="single+peer+ordered", fixture=f, eps=e, form=fm) for f, e, fm in CASES[world]])
        return cache[world]
    return get


@pytest.mark.parametrize("world,case", [(8, 0), (8, 1), (4, 0), (4, 1)])
def test_four_and_eight_ranks_on_one_gpu(runs, world, case):
    """4 and 8 processes on the one GPU, both forms of the sharded PCG, two transports in the same processes: the peer-mapped mailboxes (PEER_MAX = 8 ranks,
    dev_peer.h) and the host-staged callback adding in rank order.  'gen:odd' has odd m and n, no multiples of 8 * 32: the exchange's chunks (peer_chunk,
    rounded to 32) do not tile the vectors and the last ranks' chunks are short.  Asserted: every rank holds the same bits, the two transports agree BIT
    for bit, the row blocks tile [0, m), and the counts are the single-GPU solve's (rank 0 runs it first, on its own) = the reference's where a fixture exists."""
    name, eps, form = CASES[world][case]
    out = runs(world)[case]
    a, b, one = out, out["second"], out["single"]
    assert out["fixture"] == name and out["form"] == form
    if name.startswith("gen:"):
        m = out["shape"][0]
        assert m % 2 == 1 and out["shape"][1] % 2 == 1
    else:
        z, A, bb, c = load(name)
        g = info_of(z, f"indirect_{eps:g}")
        assert (one["ipm_iter"], one["admm_iter"]) == (g["ipm_iter"], g["admm_iter"])      # the one-GPU solve takes the reference's counts ...
        for k in "xys":
            assert rel(np.array(one[k]), z[f"indirect_{eps:g}_{k}"]) < 10 * eps, k
        m = A.shape[0]
    for r in (a, b):
        assert r["consistent"] and r["status"] == one["status"] == "Solved" and r["cols"] == (1.0 if form == "cols" else 0.0)
        assert (r["ipm_iter"], r["admm_iter"]) == (one["ipm_iter"], one["admm_iter"]), (r["transport"], r["ipm_iter"], r["admm_iter"], one["ipm_iter"], one["admm_iter"])  # ... and so do 4 / 8 ranks
        rows = r["rank_rows"]
        assert len(rows) == world and rows[0][0] == 0 and rows[-1][1] == m and all(rows[q][1] == rows[q + 1][0] and rows[q][1] > rows[q][0] for q in range(world - 1))
        for k in "xys":
            assert rel(np.array(r[k]), np.array(one[k])) < 10 * eps, k
    assert a["cg"] == b["cg"] and a["pobj"] == b["pobj"]
    for k in "xys":
        assert np.array_equal(np.array(a[k]), np.array(b[k])), k          # the two transports: the same bits


QUICK = ["--workload", "c3", "--steps", "1", "--warmup", "1", "--no-to-tol", "--no-cpu", "--no-extra"]


@pytest.mark.parametrize("transport", ["gloo-callback"])   # (the mailboxes with eight ranks: the test above; bench.py over them: scripts/r06_eight_rank.sh, profiles/r06_eight_rank_dry_run.txt)
def test_eight_ranks_started_by_bench_itself(built, transport):
    """bench.py --gpus 8 as the driver will start it on an eight-GPU node, here with every rank on cuda:0 (a functional dry run, the line says so): the
    8-way partitioner, spawn_ranks with eight children, the collectives counted; over the host-staged transport both forms of the sharded PCG on one line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ABIP_BENCH_TRANSPORT=transport, **({"ABIP_BENCH_ONE_FORM": "1"} if transport == "peer" else {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + QUICK, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["rccl_ranks"] == 8 and "NOT a scaling number" in r["transport"]
    rows = r["rank_rows"]
    assert len(rows) == 8 and rows[0][0] == 0 and rows[-1][1] == r["extra"]["m"] and all(rows[q][1] == rows[q + 1][0] and rows[q][1] > rows[q][0] for q in range(7))
    assert r["steps"] == 1 and r["value"] > 0 and r["dist_cg"] == "cols" and r["extra"]["setup_wall_s"] > 0
    forms = [r["extra"]["collectives"]]
    if transport != "peer":
        assert r["extra"]["dist_rows"]["value"] > 0
        assert abs(r["extra"]["cg_iters_per_step"] - r["extra"]["dist_rows"]["cg_iters_per_step"]) <= 0.02 * r["extra"]["cg_iters_per_step"] + 1   # the same trajectory (the forms add in different orders: a PCG count may move by one)
        forms.append(r["extra"]["dist_rows"]["collectives"])
    for c in forms:
        assert c["collectives_per_step"] > 1 and c["bytes_per_step"] > 0
