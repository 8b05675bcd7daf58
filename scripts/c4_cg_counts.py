"""PCG iterations per ADMM iteration on C4, step by step: how far does the count move from one iteration to the next (what the number of PCG iterations enqueued
blind -- next_chunk -- has to cover)?    python scripts/c4_cg_counts.py [steps]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abip_amd import Solver, problems
A, b, c = problems.lp_random_sparse()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
with Solver(A, b, c, linsys="indirect", eps=1e-6, verbose=0) as S:
    S.begin()
    prev_tot, prev, hist, seq = 0, None, collections.Counter(), []
    for k in range(N):
        S.step(1)
        tot = int(S.scalar("tot_cg_its")); its = tot - prev_tot; prev_tot = tot
        seq.append(its)
        if prev is not None:
            hist[its - prev] += 1
        prev = its
print("first 120 counts (an outer iteration's search adds its solves' counts to the step that follows it):", seq[:120])
print("change from one step to the next:", sorted(hist.items()))
