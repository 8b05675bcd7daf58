"""GPU box: a short slice of the c2 / c3 trajectory (persistent launch, dev_xcd.h) for `rocprofv3 --pmc ... -- python3 scripts/pmc_lp.py <c2|c3> [steps]`.
Prints the number of inner iterations that ran, so that the counter sums of the k_lp_xcd dispatches can be put per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from abip_amd import Solver

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
A, b, c, linsys, desc = bench.make_workload(name)
S = Solver(A, b, c, linsys=linsys, eps=1e-6, verbose=0)
S.begin()
fin, done = S.step(steps)
S.sync()
print("PMC_LP", name, "iterations", done, "workgroups", int(S.scalar("xcd_g")), flush=True)
S.close()
