"""GPU: what the plug-in boundary (include/abip_linsys.h) costs and buys -- the reference's own CPU loop with (a) its own linsys/direct.c or indirect.c,
(b) libabip_hip_linsys.so behind the same calls (every solve and SpMV crosses PCIe), against (c) the whole loop on the device (abip_init / abip_solve of
libabip_hip.so).  Same LP, same eps; iteration counts and objectives printed to show the three follow the same path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
from abip_amd import Solver, problems
from oracle import pyoracle as po

cases = [("multicommodity 120/520/8", problems.lp_multicommodity(), 1e-4), ("multicommodity 400/1500/10", problems.lp_multicommodity(nodes=400, arcs=1500, commodities=10), 1e-4),
         ("multicommodity 800/3000/10", problems.lp_multicommodity(nodes=800, arcs=3000, commodities=10), 1e-4)]  # (the reference's own LDL' on a 20 000 x 50 000 random LP: > 15 min, dropped)
for tag, (A, b, c), eps in cases[: int(sys.argv[1]) if len(sys.argv) > 1 else 2]:
    print(f"{tag}: m {A.shape[0]} n {A.shape[1]} nnz {A.nnz} eps {eps:g}", flush=True)
    for linsys in ("direct", "indirect"):
        os.environ["ABIP_HIP_LINSYS"] = linsys
        rows = []
        for label, fn in (("reference loop + its own %s.c (1 core)" % linsys, lambda: po.solve("ref", A, b, c, linsys=linsys, eps=eps, verbose=0).info),
                          ("reference loop + libabip_hip_linsys.so", lambda: po.solve("ref", A, b, c, linsys="hiplinsys", eps=eps, verbose=0).info)):
            t = time.time(); info = fn(); t = time.time() - t
            rows.append((label, info, t))
        t = time.time()
        with Solver(A, b, c, linsys=linsys, eps=eps, verbose=0) as S:
            info = S.solve()
        rows.append(("whole loop on the device (libabip_hip.so)", info, time.time() - t))
        for label, info, t in rows:
            st, sv = info.get("setup_time", 0) / 1e3, info.get("solve_time", 0) / 1e3
            print(f"  {linsys:8s} {label:48s} status {info['status_val']} ipm {info['ipm_iter']} admm {info['admm_iter']} pobj {info['pobj']:.8g} setup {st:.3f}s solve {sv:.3f}s "
                  f"({info['admm_iter'] / max(sv, 1e-9):.0f} it/s) wall {t:.2f}s", flush=True)
