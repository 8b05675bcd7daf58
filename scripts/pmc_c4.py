"""GPU box: a short slice of the C4 trajectory for `rocprofv3 --pmc ... -- python3 scripts/pmc_c4.py [steps]` (counter passes need
few dispatches: 3 warm-up + `steps` ADMM iterations of the default bench workload, nothing else)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abip_amd import Solver, problems

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
A, b, c = problems.lp_random_sparse(m=200_000, n=500_000, per_col=16)
S = Solver(A, b, c, linsys="indirect", eps=1e-6, verbose=0)
S.begin()
S.step(3 + steps)
S.sync()
S.close()
