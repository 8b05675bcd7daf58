"""Per-kernel HBM traffic from the counter passes of scripts/r03_pmc.sh: median FETCH_SIZE / WRITE_SIZE (KB) per dispatch of every kernel of interest,
traffic_bytes = 2 x FETCH + WRITE (gfx950 tallies a 128-byte read request as 64 bytes: MI355X_MICROARCH.md, HBM).  Launches that returned at a gate
(PCG launches enqueued past convergence: a few KB) are dropped: only dispatches above 5 % of the kernel's largest one count."""
import csv, glob, json, os, statistics, sys

out = sys.argv[1]
want = ("k_tail_sym", "k_tail_sym_fin", "k_tri_wide", "k_tri_wide_lds", "k_tail_mv", "kq_pcg_Gp", "kq_pcg_Aty_lds", "kq_pcg_Aty", "kq_pcg_update", "k_cg_init_A", "k_cg_init_At",
        "k_cg_spmv_A", "k_cg_spmv_At", "k_post_At", "k_q_both", "kq_rhs", "kq_ut_prox", "kq_cones", "kq_inner_both", "kq_resid", "kq_resid_lasso")
res = {}
for case in ("c4", "c5_direct", "c5_pcg", "lasso_pcg"):
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(out, f"{case}_{ctr}", "*", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] != ctr:
                    continue
                nm = row["Kernel_Name"]
                short = nm.split("(")[0].split("::")[-1].split("<")[0].strip()
                if short.startswith("void "):
                    short = short[5:]
                if short in want:
                    vals.setdefault(short, {}).setdefault(ctr, []).append(float(row["Counter_Value"]))
    rec = {}
    for k, d in sorted(vals.items()):
        def med(v):
            if not v:
                return 0.0
            big = max(v)
            w = [x for x in v if x > 0.05 * big] or v
            return statistics.median(w)
        fk, wk = med(d.get("FETCH_SIZE", [])), med(d.get("WRITE_SIZE", []))
        rec[k] = dict(traffic_bytes=int(1024 * (2 * fk + wk)), fetch_kb_median=fk, write_kb_median=wk, dispatches=len(d.get("FETCH_SIZE", [])))
    res[case] = rec
# the persistent launch (c2 / c3): one kernel, batches of iterations per dispatch -> counter SUMS over its dispatches per inner iteration
# (the solve-only launches of the Barzilai-Borwein search are k_lp_xcd dispatches too: their traffic is part of what an iteration costs)
import re
for case in ("c2", "c3"):
    tot, its, wg = {}, None, None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(out, f"{case}_{ctr}", "*", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == ctr and "k_lp_xcd" in row["Kernel_Name"]:
                    tot[ctr] = tot.get(ctr, 0.0) + float(row["Counter_Value"])
                    tot["n_" + ctr] = tot.get("n_" + ctr, 0) + 1
        log = os.path.join(out, f"{case}_{ctr}.log")
        if os.path.exists(log):
            m = re.search(r"PMC_LP \S+ iterations (\d+) workgroups (\d+)", open(log).read())
            if m:
                its, wg = int(m.group(1)), int(m.group(2))
    if its and "FETCH_SIZE" in tot:
        fk, wk = tot["FETCH_SIZE"], tot.get("WRITE_SIZE", 0.0)
        res[case] = dict(k_lp_xcd=dict(traffic_bytes_per_iteration=int(1024 * (2 * fk + wk) / its), fetch_kb_total=fk, write_kb_total=wk, iterations=its,
                                       dispatches=tot.get("n_FETCH_SIZE", 0), workgroups=wg))
json.dump(res, sys.stdout, indent=1)
