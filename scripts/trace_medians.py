"""From a rocprofv3 --kernel-trace CSV: per kernel, launches, total, median, and median / mean of the launches that did work (>= 8 us: a PCG launch
enqueued past convergence returns at its gate in 2-4 us).  With --json OUT and --bench LINE.json also writes the durations file bench.py reads
(profiles/r04_trace_durations.json): the trace duration of the C4 SpMV kernels next to the device-stamp figure of the SAME profiled run."""
import csv, json, statistics, sys
from collections import defaultdict

def short(name):
    name = name.split("(")[0]
    for pre in ("void abip::", "abip::", "void "):
        if name.startswith(pre): name = name[len(pre):]
    return name

def main():
    args = sys.argv[1:]
    trace = args[0]
    jout = args[args.index("--json") + 1] if "--json" in args else None
    bench = args[args.index("--bench") + 1] if "--bench" in args else None
    cmd = args[args.index("--cmd") + 1] if "--cmd" in args else ""
    d = defaultdict(list)
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# from the kernel trace of:", cmd)
    rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
    for k, v in rows:
        w = [x for x in v if x >= 8.0]
        print("%-44s launches %6d  total %10.1f us  median(all) %7.2f  median(working, >= 8 us) %7.2f  mean(working) %7.2f  n_working %d" % (
            k[:44], len(v), sum(v), statistics.median(v), statistics.median(w) if w else 0.0, statistics.mean(w) if w else 0.0, len(w)))
    if jout and bench:
        line = json.load(open(bench))
        roof = line["roofline"]
        stamp = {"k_cg_spmv_A": None, "k_cg_spmv_At": None}
        side = roof.get("stamps") or roof
        main_k = "k_cg_spmv_A" if "spmv_A " in roof["kernel"] or "k_cg_spmv_A " in roof["kernel"] else "k_cg_spmv_At"
        other_k = "k_cg_spmv_At" if main_k == "k_cg_spmv_A" else "k_cg_spmv_A"
        stamp[main_k] = side["avg_launch_us"]
        oth = list((roof.get("other_spmv") or {}).values())
        if oth: stamp[other_k] = oth[0].get("stamp_avg_launch_us", oth[0]["avg_launch_us"])
        out = {}
        for k, v in d.items():
            base = k.split("<")[0]
            if base in stamp and stamp[base]:
                w = [x for x in v if x >= 8.0]
                out[base] = dict(median_working_us=statistics.median(w), mean_working_us=statistics.mean(w), n_working=len(w), launches=len(v), stamp_avg_us_same_run=stamp[base])
        out["source"] = "kernel trace of: " + cmd
        # which kernels this ratio was measured on: bench.py ignores the file (and says so) when the sources have changed since
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from abip_amd import _lib
        out["kernel_sources_sha256"] = _lib.kernel_sources_sha256()   # (one helper with bench.py: dev_kernels.h, every header it includes, solver.hip)
        out["note"] = ("kernel-trace duration (dispatch to drain) of the launches that did work, and the device-side stamp figure bench.py measured in the SAME profiled run: "
                       "the stamps (first sampled workgroup begin .. last sampled workgroup end) leave out dispatch and drain")
        json.dump(out, open(jout, "w"), indent=1)

if __name__ == "__main__":
    main()
