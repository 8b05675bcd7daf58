"""GPU: soak of the persistent launch -- repeated full solves of the c2 / c3 surrogates (and of c3 with the workgroup count forced), every run must
reproduce the first one bit for bit; prints the spread of the solve times.   usage: xcd_soak.py [repeats_c2] [repeats_c3]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from abip_amd import Solver

def soak(name, reps, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    A, b, c, linsys, desc = bench.make_workload(name)
    first, times = None, []
    for r in range(reps):
        with Solver(A, b, c, linsys=linsys, eps=1e-6, verbose=0) as S:
            t0 = time.perf_counter(); info = S.solve(); times.append(time.perf_counter() - t0)
            h = (info["admm_iter"], hashlib.md5(np.concatenate([S.x, S.y, S.s]).tobytes()).hexdigest())
            g = int(S.scalar("xcd_g"))
        if first is None:
            first = h
        assert h == first, (name, env, r, h, first)
    for k in (env or {}):
        os.environ.pop(k, None)
    os.write(bench._REAL_STDOUT, ("ok  %-3s %-22s %3d solves, workgroups %3d, %6d iterations each, identical bits; solve time min %.3f median %.3f max %.3f s\n" % (
        name, str(env or ""), reps, g, first[0], min(times), sorted(times)[len(times) // 2], max(times))).encode())   # (importing bench points the C-level stdout at stderr)

n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n3 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
soak("c2", n2)
soak("c3", n3)
for G in ("32", "64", "256"):
    soak("c3", 2, {"ABIP_HIP_XCD_G": G})
