"""GPU: the streamed launch path on C4, repeated: every whole solve must reproduce the first one bit for bit (iterations, PCG total, x / y / s) whatever the
stalls did; prints the spread of the solve times.    python scripts/c4_soak.py [repeats]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from abip_amd import Solver, problems
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
A, b, c = problems.lp_random_sparse()
first, times, stalls = None, [], []
for r in range(reps):
    with Solver(A, b, c, linsys="indirect", eps=1e-6, verbose=0) as S:
        info = S.solve(); times.append(info["solve_time"] / 1e3); stalls.append(int(S.scalar("stream_stalls")))
        h = (info["ipm_iter"], info["admm_iter"], int(S.scalar("tot_cg_its")), hashlib.md5(np.concatenate([S.x, S.y, S.s]).tobytes()).hexdigest())
    if first is None:
        first = h
    assert h == first, (r, h, first)
os.write(bench._REAL_STDOUT, ("ok  C4 %d whole solves to eps 1e-6: %d / %d iterations, %d PCG iterations, identical bits; solve time min %.2f median %.2f max %.2f s; stalls per solve %s\n" % (
    reps, first[0], first[1], first[2], min(times), sorted(times)[len(times) // 2], max(times), stalls)).encode())
