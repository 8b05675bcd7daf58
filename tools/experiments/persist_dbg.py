import os, sys, subprocess
ROOT = "/root/repo" if os.path.isdir("/root/repo/abip_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from _golden import load
import abip_amd
name, mode = sys.argv[1], sys.argv[2]
z, A, b, c = load(name)
with abip_amd.Solver(A, b, c, linsys="direct", verbose=0, eps=1e-6) as S:
    print(name, mode, "persist", S.scalar("persist"), "tail", S.scalar("tail"), "nb", S.scalar("nb"), flush=True)
    if mode == "strided":
        S.begin(); fin = False
        while not fin: fin, done = S.step(13)
        info = S.end()
    else:
        info = S.solve()
    print("   ->", info["status"], info["ipm_iter"], info["admm_iter"], flush=True)
''' % (ROOT, ROOT)
for name in ("lp_afiro_like", "lp_staircase", "lp_random_sparse_small"):
    for mode, env in (("persist", {"ABIP_HIP_PERSIST": "1"}), ("stepwise", {"ABIP_HIP_PERSIST": "1", "ABIP_HIP_BATCH": "0"}), ("strided", {"ABIP_HIP_PERSIST": "1"}),
                      ("wgs5", {"ABIP_HIP_PERSIST": "1", "ABIP_HIP_PERSIST_WGS": "5"}), ("launches", {})):
        e = dict(os.environ); e.pop("ABIP_HIP_PERSIST", None); e.update(env)
        try:
            p = subprocess.run([sys.executable, "-c", code, name, mode], env=e, capture_output=True, text=True, timeout=90)
            print([l for l in p.stdout.splitlines() if not "rescaling" in l], "rc", p.returncode, flush=True)
        except subprocess.TimeoutExpired as ex:
            print(name, mode, "TIMEOUT", (ex.stdout or b"")[-300:], flush=True)
