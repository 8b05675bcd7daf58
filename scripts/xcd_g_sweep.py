"""GPU: the persistent launch's workgroup count (ABIP_HIP_XCD_G = 32 / 64 / 128 / 256: 1 / 2 / 4 / 8 XCDs) against the non-zero count, PCG back-end -- multicommodity
network LPs of growing size and two random sparse ones; fixed window of ADMM iterations each.  Where the planner's threshold (solver.hip: xcd_plan) comes from.
usage: xcd_g_sweep.py [steps] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from abip_amd import Solver, problems

if "--big" in sys.argv:
    pass
steps = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 150
warm = int([a for a in sys.argv[1:] if a.isdigit()][1]) if len([a for a in sys.argv[1:] if a.isdigit()]) > 1 else 30
cases = [("multicommodity 60 nodes / 260 arcs / 8", lambda: problems.lp_multicommodity(nodes=60, arcs=260, commodities=8)[:3]),
         ("multicommodity 120 / 520 / 8 (the small fixture's generator)", lambda: problems.lp_multicommodity()[:3]),
         ("multicommodity 240 / 1100 / 12", lambda: problems.lp_multicommodity(nodes=240, arcs=1100, commodities=12)[:3]),
         ("multicommodity 400 / 2000 / 16", lambda: problems.lp_multicommodity(nodes=400, arcs=2000, commodities=16)[:3]),
         ("c3 (bench)", lambda: bench.make_workload("c3")[:3]),
         ("multicommodity 900 / 4400 / 24", lambda: problems.lp_multicommodity(nodes=900, arcs=4400, commodities=24)[:3]),
         ("random sparse 8000 x 20000 x 4", lambda: problems.lp_random_sparse(m=8000, n=20000, per_col=4, seed=7)[:3]),
         ("random sparse 20000 x 50000 x 4", lambda: problems.lp_random_sparse(m=20000, n=50000, per_col=4, seed=8)[:3]),
         ("random sparse 20000 x 50000 x 8", lambda: problems.lp_random_sparse(m=20000, n=50000, per_col=8, seed=9)[:3]),
         ("random sparse 20000 x 50000 x 16 (test_gpu_xcd.py)", lambda: problems.lp_random_sparse(m=20000, n=50000, per_col=16, seed=3)[:3])]
if "--big" in sys.argv:
    cases = cases[5:6] + cases[7:]
    sys.argv.remove("--big")
def emit(s):
    os.write(bench._REAL_STDOUT, (s + "\n").encode())
for name, make in cases:
    A, b, c = make()
    row = []
    for g in (32, 64, 128, 256):
        os.environ["ABIP_HIP_XCD_G"] = str(g)
        best = 0.0
        try:
            for rep in range(2):
                with Solver(A, b, c, linsys="indirect", eps=1e-9, verbose=0) as S:
                    if S.scalar("xcd") != 1.0:
                        best = -1.0
                        break
                    S.begin(); S.step(warm); S.sync()
                    t0 = time.perf_counter(); fin, done = S.step(steps); S.sync(); dt = time.perf_counter() - t0
                    best = max(best, done / dt)
        except Exception as e:  # a plan that does not fit
            best = -1.0
        row.append(best)
    os.environ.pop("ABIP_HIP_XCD_G")
    with Solver(A, b, c, linsys="indirect", eps=1e-9, verbose=0) as S:
        chosen = int(S.scalar("xcd_g")) if S.scalar("xcd") == 1.0 else 0
    os.environ["ABIP_HIP_XCD"] = "0"
    lp = 0.0
    for rep in range(2):
        with Solver(A, b, c, linsys="indirect", eps=1e-9, verbose=0) as S:
            S.begin(); S.step(warm); S.sync()
            t0 = time.perf_counter(); fin, done = S.step(steps); S.sync(); dt = time.perf_counter() - t0
            lp = max(lp, done / dt)
    os.environ.pop("ABIP_HIP_XCD")
    emit("%-62s %6d x %6d nnz %7d | G 32: %8.0f  64: %8.0f  128: %8.0f  256: %8.0f it/s | launch path %8.0f | planner: %d" % (name, A.shape[0], A.shape[1], A.nnz, *row, lp, chosen))
