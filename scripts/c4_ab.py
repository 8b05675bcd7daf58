"""A / B of the launch path on C4 (BASELINE configs[3]) under environment switches: ABIP_HIP_ATY (the back-substitution's A'u_y kept for the stopping test and the next
solve's set-up), ABIP_HIP_STREAM (iterations streamed: the host one verdict behind the device), ABIP_HIP_STREAM_BB (the Barzilai-Borwein search streamed too, its decisions on the device),
ABIP_HIP_BB_REUSE (a look-ahead whose penalty did not change hands its second step to the next one instead of solving the same system again).  Prints the driver's window (20 steps after 5), a 200-step window and,
with --solve, the whole solve to eps 1e-6; checks that every variant leaves the same iterate.

    python scripts/c4_ab.py [--solve] [--out FILE]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from abip_amd import Solver, problems

OUT = open(sys.argv[sys.argv.index("--out") + 1], "w") if "--out" in sys.argv else sys.stdout
def emit(*a):
    print(*a, file=OUT, flush=True)

A, b, c = problems.lp_random_sparse()
VARIANTS = [("round 4 form", {"ABIP_HIP_ATY": "0", "ABIP_HIP_STREAM": "0", "ABIP_HIP_STREAM_BB": "0"}), ("A'u_y kept", {"ABIP_HIP_ATY": "1", "ABIP_HIP_STREAM": "0", "ABIP_HIP_STREAM_BB": "0"}),
            ("+ streamed", {"ABIP_HIP_ATY": "1", "ABIP_HIP_STREAM": "1", "ABIP_HIP_STREAM_BB": "0"}), ("+ search", {"ABIP_HIP_ATY": "1", "ABIP_HIP_STREAM": "1", "ABIP_HIP_STREAM_BB": "1", "ABIP_HIP_BB_REUSE": "0"}),
            ("+ reuse (default)", {"ABIP_HIP_ATY": "1", "ABIP_HIP_STREAM": "1", "ABIP_HIP_STREAM_BB": "1", "ABIP_HIP_BB_REUSE": "1"})]
if "--quick" in sys.argv:
    VARIANTS = VARIANTS[2:]
ref = None
for name, env in VARIANTS:
    os.environ.update(env)
    with Solver(A, b, c, linsys="indirect", eps=1e-6, verbose=0) as S:
        S.begin()
        S.step(5); S.sync()
        t0 = time.perf_counter(); S.step(20); S.sync(); t20 = time.perf_counter() - t0
        t0 = time.perf_counter(); S.step(200); S.sync(); t200 = time.perf_counter() - t0
        u = S.vector("u").copy(); cg = S.scalar("tot_cg_its")
        same = "-" if ref is None else ("same bits" if np.array_equal(u, ref) else "rel diff %.1e" % (np.linalg.norm(u - ref) / np.linalg.norm(ref)))
        if ref is None:
            ref = u
        emit("%-20s  20-step window %7.1f it/s   200-step window %7.1f it/s   PCG iterations so far %d   u after 225 iterations: %s   stalls %d" % (name, 20 / t20, 200 / t200, cg, same, S.scalar("stream_stalls")))
    if "--solve" in sys.argv:
        with Solver(A, b, c, linsys="indirect", eps=1e-6, verbose=0) as S:
            t0 = time.perf_counter(); info = S.solve(); t = time.perf_counter() - t0
            emit("%-20s  whole solve: %s  %d / %d iterations  %.2f s  %.1f it/s  pobj %.12g  PCG %d  stalls %d" % (name, info["status"], info["ipm_iter"], info["admm_iter"], info["solve_time"] / 1e3,
                 info["admm_iter"] / (info["solve_time"] / 1e3), info["pobj"], S.scalar("tot_cg_its"), S.scalar("stream_stalls")))
