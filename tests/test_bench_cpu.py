"""CPU: bench.py refuses an N-GPU launch it cannot make (no GPUs here) before touching a GPU -- non-zero exit, no JSON line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_over=None):
    env = dict(os.environ, **(env_over or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        if not env_over or k not in env_over:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_more_gpus_than_the_node_has_is_refused():
    import torch
    if torch.cuda.device_count() >= 8:
        return
    p = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert p.returncode != 0
    assert "needs 8 GPUs" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_world_size_contradicting_gpus_is_refused():
    p = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "contradicts WORLD_SIZE" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
