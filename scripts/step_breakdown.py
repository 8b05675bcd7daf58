#!/usr/bin/env python3
"""Where the wall time of an ADMM step goes: reads a rocprofv3 kernel trace (…_kernel_trace.csv) of `bench.py --workload c4 --steps K --warmup W`
and breaks the window of the LAST K steps down into (a) kernels that did work, (b) launches that found nothing to do (chunk launches enqueued past
the PCG's convergence: they read the control block and return), (c) gaps between launches by size.   usage: step_breakdown.py <trace.csv> [K=20]"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("abip::", "")
    return name.split("(")[0]


def main():
    path, K = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    upd = [i for i, r in enumerate(rows) if r[2] == "k_admm_update" and r[1] - r[0] >= 3000]
    if len(upd) < K + 1:
        print(f"only {len(upd)} working k_admm_update launches in the trace")
        return
    i0, i1 = upd[-K - 1] + 1, upd[-1]      # from the launch after step -K-1's update through step -1's update
    win = rows[i0:i1 + 1]
    t0, t1 = rows[i0 - 1][1], win[-1][1]
    wall = (t1 - t0) / 1e3
    work, noop = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    gaps = defaultdict(lambda: [0, 0.0])
    edges = [(0, 1), (1, 2), (2, 5), (5, 10), (10, 50), (50, 1e9)]
    prev_end = t0
    busy = 0.0
    after = defaultdict(lambda: [0, 0.0])
    prev_name = rows[i0 - 1][2]
    for s, e, nm in win:
        d = (e - s) / 1e3
        tgt = noop if d < 3.0 else work
        tgt[nm][0] += 1
        tgt[nm][1] += d
        g = (s - prev_end) / 1e3
        if g > 0:
            for lo, hi in edges:
                if lo <= g < hi:
                    gaps[(lo, hi)][0] += 1
                    gaps[(lo, hi)][1] += g
            if g >= 5:
                after[prev_name + " -> " + nm][0] += 1
                after[prev_name + " -> " + nm][1] += g
        busy += (min(e, t1) - max(s, prev_end)) / 1e3 if e > prev_end else 0.0
        prev_end = max(prev_end, e)
        prev_name = nm
    tw = sum(v[1] for v in work.values())
    tn = sum(v[1] for v in noop.values())
    tg = sum(v[1] for v in gaps.values())
    print(f"window: the last {K} ADMM steps of the trace, {len(win)} launches, wall {wall:.1f} us = {wall / K:.1f} us per step -> {1e6 * K / wall:.1f} it/s inside the profiler")
    print(f"  kernels that did work   {tw:10.1f} us  {100 * tw / wall:5.1f} %")
    print(f"  launches with no work   {tn:10.1f} us  {100 * tn / wall:5.1f} %   ({sum(v[0] for v in noop.values())} launches)")
    print(f"  gaps between launches   {tg:10.1f} us  {100 * tg / wall:5.1f} %   ({sum(v[0] for v in gaps.values())} gaps)")
    print("\nkernels that did work (>= 3 us):")
    for nm, (c, t) in sorted(work.items(), key=lambda kv: -kv[1][1]):
        print(f"  {nm:44s} {c:6d} launches {t:10.1f} us  {100 * t / wall:5.1f} %  avg {t / c:7.2f} us")
    print("\nlaunches that found nothing to do (< 3 us):")
    for nm, (c, t) in sorted(noop.items(), key=lambda kv: -kv[1][1]):
        print(f"  {nm:44s} {c:6d} launches {t:10.1f} us  {100 * t / wall:5.1f} %  avg {t / c:7.2f} us")
    print("\ngaps between the end of one launch and the start of the next:")
    for (lo, hi), (c, t) in sorted(gaps.items()):
        print(f"  [{lo:>3g}, {hi:>5g}) us {c:6d} gaps {t:10.1f} us  {100 * t / wall:5.1f} %")
    print("\ngaps >= 5 us by the launches either side (host round trips: the chunked PCG reads its count back, the step boundary):")
    for nm, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {nm:70s} {c:5d} gaps {t:9.1f} us  avg {t / c:7.1f} us")


if __name__ == "__main__":
    main()
