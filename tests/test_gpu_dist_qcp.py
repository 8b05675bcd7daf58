"""GPU (-m gpu): the conic path with its columns sharded over several ranks (abip_amd/csrc/qcp_dist.h), on ONE GPU through the host-staged
gloo collective: same iteration counts and solution as the single-GPU run, every rank bit-identical."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from qcp_cases import make

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
    from abip_amd import qcp
    return qcp


def rel(a, r):
    a, r = np.asarray(a), np.asarray(r)
    return np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-300)


@pytest.mark.parametrize("name", ["lasso_small", "mixed", "lp"])
def test_one_rank_sharded_conic_path_equals_plain_path(gpu, name):
    """world = 1 through the sharded code path (split products + exchange + element-wise halves) with an identity collective."""
    from abip_amd import dist as adist
    data, K = make(name)
    stg = dict(eps=1e-3 if name == "lp" else 1e-5, linsys_solver=3, verbose=0)   # (the LP case needs ~50 000 iterations at 1e-5: scripts/gpu_sweep_dist_qcp.sh runs it)
    ref, ri = gpu.abip_qcp(data, K, stg)
    adist.init_callback(0, 1, lambda arr: None)
    try:
        got, gi = gpu.abip_qcp(data, K, stg)
    finally:
        adist.finalize()
    assert gi["status"] == ri["status"] == "Solved"
    assert gi["ipm_iter"] == ri["ipm_iter"] and abs(gi["admm_iter"] - ri["admm_iter"]) <= 2
    for k in "xys":
        assert rel(got[k], ref[k]) < 1e-6, k
    assert gi["factor"]["head_nnz"] > gi["admm_iter"]          # (sharded runs report the collectives issued there)


@pytest.mark.parametrize("world,name", [(2, "lasso_small"), (2, "mixed"), (3, "mixed"), (3, "lp")])   # (scripts/gpu_sweep_dist_qcp.sh runs the wider matrix)
def test_multi_rank_conic_sharding(gpu, world, name):
    data, K = make(name)
    eps = 1e-3 if name == "lp" else 1e-5
    ref, ri = gpu.abip_qcp(data, K, dict(eps=eps, linsys_solver=3, verbose=0))
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_qcp.py"), "gloo-callback", name, repr(eps)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert p.returncode == 0 and lines, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(lines[-1][7:])
    assert out["consistent"] and out["status"] == ri["status"] == "Solved"
    assert out["ipm_iter"] == ri["ipm_iter"] and abs(out["admm_iter"] - ri["admm_iter"]) <= 0.03 * ri["admm_iter"] + 3
    for k in "xys":
        assert rel(out[k], ref[k]) < 30 * eps, k
    assert abs(out["pobj"] - ri["pobj"]) <= 30 * eps * (1 + abs(ri["pobj"]))


@pytest.mark.parametrize("world,pt,cname", [(2, 0, "wide_sparse"), (3, 3, "tall")])
def test_multi_rank_ml_front_ends(gpu, world, pt, cname):
    """The LASSO and SVM-QP front ends shard like the generic formulation (their operators are materialised matrices; the LASSO residual kernel adds
    only the columns the rank owns).  The SVM-SOCP front end stays a replica: its residuals pair entries of different columns."""
    eps = 1e-5
    if pt == 0:
        from _lasso_cases import gen
        X, yv, lam = gen(cname)
    else:
        from _svm_cases import gen
        X, yv = gen(cname)
        lam = 1e-2
    ref, ri = gpu.abip_ml(dict(X=X, y=yv, **{"lambda": lam}), dict(prob_type=pt, eps=eps, linsys_solver=3, verbose=0))
    port = 29450 + world + pt
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_qcp.py"), "gloo-callback", f"ml:{pt}:{cname}", repr(eps)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert p.returncode == 0 and lines, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(lines[-1][7:])
    assert out["consistent"] and out["status"] == ri["status"] == "Solved" and out["collectives"] > out["admm_iter"]
    assert out["ipm_iter"] == ri["ipm_iter"] and abs(out["admm_iter"] - ri["admm_iter"]) <= 0.03 * ri["admm_iter"] + 3
    assert rel(out["x"], ref["x"]) < 30 * eps
    assert abs(out["pobj"] - ri["pobj"]) <= 30 * eps * (1 + abs(ri["pobj"]))


def test_sharded_conic_time_limit(gpu):
    """A finite time limit in a sharded solve: the ranks sum their verdicts (one small collective per test) and stop together; here one rank, limit already
    exceeded at the first test -> the solve hands back what it has with an inaccurate status instead of running on."""
    from abip_amd import dist as adist
    data, K = make("lasso_small")
    adist.init_callback(0, 1, lambda arr: None)
    try:
        sol, info = gpu.abip_qcp(data, K, dict(eps=1e-9, linsys_solver=3, verbose=0, time_limit=1e-6))
    finally:
        adist.finalize()
    assert info["status"] == "Solved/Inaccurate" and info["admm_iter"] <= 3 and np.all(np.isfinite(sol["x"]))
