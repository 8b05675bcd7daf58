"""GPU (one device): the sharded PCG path with 2 and 3 ranks (host-staged gloo collective, every rank on cuda:0) on seeded LPs,
against the single-rank device run of the same problem: same status / outer count, inner count within 3 %, ranks consistent."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

def run(world, fixture, eps, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), "gloo-callback", fixture, repr(eps)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    if p.returncode != 0 or not lines:
        return None, (p.stdout[-500:] + p.stderr[-800:])
    return json.loads(lines[-1][7:]), ""

bad = 0
port = 29600
for kind in ("rand", "stair", "mc"):
    for seed in (int(sys.argv[1]) if len(sys.argv) > 1 else 1, (int(sys.argv[1]) if len(sys.argv) > 1 else 1) + 100):
        fx = f"gen:{kind}:{seed}"
        base, err = run(1, fx, 1e-5, port); port += 1
        if base is None:
            print("BAD", fx, "single-rank worker failed", err); bad += 1; continue
        for world in (2, 3):
            out, err = run(world, fx, 1e-5, port); port += 1
            if out is None:
                print("BAD", fx, world, "worker failed", err); bad += 1; continue
            rel = lambda a, r: float(np.linalg.norm(np.array(a) - np.array(r)) / max(np.linalg.norm(np.array(r)), 1e-300))
            ex = max(rel(out[k], base[k]) for k in "xy")
            ok = out["consistent"] and out["status"] == base["status"] and out["ipm_iter"] == base["ipm_iter"] and abs(out["admm_iter"] - base["admm_iter"]) <= 0.03 * base["admm_iter"] + 2
            bad += not ok
            print(f"{'ok ' if ok else 'BAD'} {fx:16s} world {world}: status {out['status']}/{base['status']} admm {out['admm_iter']}/{base['admm_iter']} ipm {out['ipm_iter']}/{base['ipm_iter']} "
                  f"cg {int(out['cg'])}/{int(base['cg'])} rel(xy) {ex:.1e} consistent {out['consistent']}", flush=True)
print("FAILURES", bad)
