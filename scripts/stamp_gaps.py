#!/usr/bin/env python3
"""Idle time between the launches of the streamed PCG path, measured OUTSIDE the profiler (rocprofv3's kernel trace showed 45-62 us gaps
inside PCG chunks; is that the profiler's doing?).

    ABIP_HIP_STAMP_DUMP=/tmp/st.txt python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu --no-extra --no-to-tol ; python scripts/stamp_gaps.py /tmp/st.txt

The two product kernels of a PCG iteration (class 0 = k_cg_spmv_At, 1 = k_cg_spmv_A) note the device wall clock (100 MHz) when their first sampled workgroup
begins and their last sampled one ends.  From consecutive records:
    At -> A      : begin(A) - end(At)                      = one launch boundary
    A -> At next : begin(At') - end(A) - (k_cg_update)     = two launch boundaries + the update kernel in between (its ~6 us are reported, not subtracted)
Records with end == 0 are launches that returned at a gate (enqueued past PCG convergence); a pair with one of those in between is a chunk boundary."""
import sys
import numpy as np

TICK_US = 0.01
rec = [tuple(int(v) for v in ln.split()) for ln in open(sys.argv[1]) if ln.strip()]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # only the last N records (the timed window)
if last:
    rec = rec[-last:]
ata, aat, dur = [], [], {0: [], 1: []}
prev = None
for cls, t0, t1 in rec:
    if t1 == 0:
        prev = None          # a no-op launch: whatever follows starts a new chain
        continue
    dur[cls].append((t1 - t0) * TICK_US)
    if prev is not None:
        pc, pt1 = prev
        g = (t0 - pt1) * TICK_US
        if pc == 0 and cls == 1:
            ata.append(g)
        elif pc == 1 and cls == 0:
            aat.append(g)
    prev = (cls, t1)
def show(name, v):
    v = np.array(v)
    if not len(v):
        print(f"{name}: none"); return
    hist = [(lo, hi, int(((v >= lo) & (v < hi)).sum())) for lo, hi in ((-1e9, 1), (1, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9))]
    print(f"{name}: {len(v)} pairs, median {np.median(v):.2f} us, mean {v.mean():.2f} us, p99 {np.percentile(v, 99):.2f} us, max {v.max():.2f} us, sum {v.sum() / 1e3:.3f} ms")
    print("    " + "  ".join(f"[{lo if lo > -1e8 else '-inf'}, {hi if hi < 1e8 else 'inf'}): {n}" for lo, hi, n in hist))
print(f"{len(rec)} stamped launches; k_cg_spmv_At: {len(dur[0])} working launches, mean {np.mean(dur[0]):.2f} us; k_cg_spmv_A: {len(dur[1])}, mean {np.mean(dur[1]):.2f} us (first sampled workgroup begin .. last sampled end)")
show("k_cg_spmv_At -> k_cg_spmv_A (one boundary)", ata)
show("k_cg_spmv_A -> [k_cg_update] -> k_cg_spmv_At (two boundaries + the update kernel)", aat)
