"""GPU: fixed-window rate of one workload through whichever libabip_hip ABIP_HIP_LIBRARY names (developer A/B of two builds on one box).
usage: xcd_ab.py <c2|c3> [steps] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from abip_amd import Solver
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else (2000 if name == "c2" else 300)
warm = int(sys.argv[3]) if len(sys.argv) > 3 else (200 if name == "c2" else 50)
A, b, c, linsys, desc = bench.make_workload(name)
best = 0.0
for rep in range(3):
    with Solver(A, b, c, linsys=linsys, eps=1e-6, verbose=0) as S:
        S.begin(); S.step(warm); S.sync()
        t0 = time.perf_counter(); fin, done = S.step(steps); S.sync(); dt = time.perf_counter() - t0
        best = max(best, done / dt)
os.write(bench._REAL_STDOUT, ("%s %s best of 3: %.0f it/s\n" % (name, os.environ.get("ABIP_HIP_LIBRARY", "default"), best)).encode())   # (importing bench points the C-level stdout at stderr)
