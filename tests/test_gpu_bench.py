"""GPU (-m gpu): bench.py's launch contract.  On the one-GPU test box:
  * the sharded code path with one rank over RCCL (ABIP_BENCH_FORCE_SHARD=1 --gpus 1),
  * --gpus 2 started BY bench.py itself as two ranks (gloo-callback dry run: both on cuda:0, host-staged sums),
  * --gpus 2 over RCCL with one GPU visible must refuse (non-zero exit, no JSON line)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--workload", "c3", "--steps", "6", "--warmup", "2", "--no-to-tol", "--no-cpu", "--no-extra"]


def _run(extra_args, env_over, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_over)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_one_rank_sharded_over_rccl():
    import torch
    assert torch.cuda.is_available()
    p, lines = _run(["--gpus", "1"] + QUICK, {"ABIP_BENCH_FORCE_SHARD": "1"})
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["scaling"] == "strong" and r["rccl_ranks"] == 1 and r["transport"] == "rccl"
    assert r["rank_rows"] == [[0, r["extra"]["m"]]] and r["value"] > 0
    assert r["roofline"]["launches"] > 0 and 0 < r["roofline"]["frac"] < 1
    assert r["dist_cg"] == "cols" and r["extra"]["collectives"]["ms_in_allreduce_per_step"] >= 0     # hipEvents around each ncclAllReduce (one rank: next to nothing)
    assert r["extra"]["collectives"]["collectives_per_step"] > 1 and r["extra"]["collectives"]["bytes_per_step"] > 0
    assert r["extra"]["dist_rows"]["collectives"]["bytes_per_step"] > 0


def test_two_ranks_started_by_bench_itself():
    p, lines = _run(["--gpus", "2"] + QUICK, {"ABIP_BENCH_TRANSPORT": "gloo-callback"})
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["rccl_ranks"] == 2 and "gloo-callback" in r["transport"]
    rows = r["rank_rows"]
    assert rows[0][0] == 0 and rows[0][1] == rows[1][0] and rows[1][1] == r["extra"]["m"] and rows[0][1] > 0
    assert r["steps"] == 6 and r["value"] > 0
    # one invocation measures both forms of the sharded solve: the headline is the library default (columns: m < n), the row form sits beside it
    assert r["dist_cg"] == "cols" and r["extra"]["dist_rows"]["value"] > 0
    for c in (r["extra"]["collectives"], r["extra"]["dist_rows"]["collectives"]):
        assert c["collectives_per_step"] > 1 and c["bytes_per_step"] > 0 and c["ms_in_allreduce_per_step"] >= 0


def test_two_ranks_over_rccl_refused_on_one_gpu():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    p, lines = _run(["--gpus", "2"] + QUICK, {})
    assert p.returncode != 0 and not lines
    assert "needs 2 GPUs" in p.stderr


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + QUICK, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert q.returncode != 0 and "contradicts WORLD_SIZE" in q.stderr and not [ln for ln in q.stdout.splitlines() if ln.startswith("{")]


def test_conic_workload_sharded_by_bench_itself():
    """--workload c5 --linsys indirect --gpus 2: the conic PCG path with its columns sharded over two ranks (gloo-callback dry run on one GPU),
    and the same through one rank over a real RCCL communicator; both reach the single-GPU iteration count."""
    p, lines = _run(["--gpus", "2", "--workload", "c5", "--linsys", "indirect", "--no-cpu"], {"ABIP_BENCH_TRANSPORT": "gloo-callback"})
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-3000:]
    r2 = json.loads(lines[0])
    assert r2["n_gpus"] == 2 and r2["scaling"] == "strong" and "sharded over 2 ranks" in r2["config"]["parallelism"] and r2["time_to_tol"]["status"] == "Solved"
    p, lines = _run(["--gpus", "1", "--workload", "c5", "--linsys", "indirect", "--no-cpu"], {"ABIP_BENCH_FORCE_SHARD": "1"})
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-3000:]
    r1 = json.loads(lines[0])
    assert "sharded over 1 ranks" in r1["config"]["parallelism"] and "1 ranks in the communicator" in r1["config"]["parallelism"]
    assert r1["steps"] == r2["steps"] == 287 and abs(r1["extra"]["pobj"] - r2["extra"]["pobj"]) < 1e-9 * abs(r1["extra"]["pobj"])
