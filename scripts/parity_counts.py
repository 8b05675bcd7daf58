"""Iteration counts of every reference fixture (tests/golden/*.npz, written by make_golden.py from oracle/_ref) on the three device paths:
the launch path (ABIP_HIP_XCD=0), the persistent launch with one batch of inner iterations per launch (ABIP_HIP_XCD_OUTER=0, round 3) and the
persistent launch that spans outer iterations (the default, round 4).  One line per (fixture, eps, back-end); `!` marks a count that differs from
the reference's.  The tolerances of tests/test_gpu_parity.py that depend on a count refer to this table (profiles/r04_parity_counts.txt)."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from _golden import TINY_VARIANTS, info_of, load, rel
import abip_amd as gpu

OUT = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout   # (the library's own chatter goes to the C-level stdout: give the table a file of its own)
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
def emit(*a):
    print(*a, file=OUT, flush=True)
MODES = (("launch path", {"ABIP_HIP_XCD": "0"}), ("persistent, batches", {"ABIP_HIP_XCD": "1", "ABIP_HIP_XCD_OUTER": "0"}), ("persistent, whole solve", {"ABIP_HIP_XCD": "1", "ABIP_HIP_XCD_OUTER": "1"}))
emit("%-26s %-9s %-8s | %-14s | %s" % ("fixture", "back-end", "eps", "reference", " | ".join("%-26s" % m[0] for m in MODES)))
bad = 0
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lp_*.npz"))):
    name = os.path.basename(f)[:-4]
    if ONLY and ONLY not in name: continue
    if name in ("lp_c4_prefix", "lp_pds_like_full"): continue   # the BASELINE-size fixtures carry no LP (tests/test_gpu_baseline_size.py rebuilds it and holds the device to them)
    z, A, b, c = load(name)
    kw = TINY_VARIANTS.get(name.replace("lp_tiny_", ""), {}) if name.startswith("lp_tiny_") else {}
    for tag in sorted(k[:-5] for k in z.keys() if k.endswith("_info")):
        linsys, eps = tag.split("_")
        g = info_of(z, tag)
        cols = []
        for mname, env in MODES:
            os.environ.update(env)
            with gpu.Solver(A, b, c, linsys=linsys, verbose=0, eps=float(eps), max_admm_iters=400000, **kw) as S:
                info = S.solve()
                d = max(rel(getattr(S, k), z[f"{tag}_{k}"]) for k in "xys")
                flag = "" if (info["ipm_iter"] == g["ipm_iter"] and info["admm_iter"] == g["admm_iter"] and info["status_val"] == g["status_val"]) else "!"
                bad += flag == "!"
                cols.append("%3d / %6d%1s  xys %.1e" % (info["ipm_iter"], info["admm_iter"], flag, d))
        emit("%-26s %-9s %-8s | %3d / %6d   | %s" % (name, linsys, eps, g["ipm_iter"], g["admm_iter"], " | ".join("%-26s" % c_ for c_ in cols)))
emit("counts that differ from the reference's:", bad)
