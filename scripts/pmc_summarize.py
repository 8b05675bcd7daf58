"""Per-kernel HBM traffic from the counter passes of scripts/r03_pmc.sh: median FETCH_SIZE / WRITE_SIZE (KB) per dispatch of every kernel of interest,
traffic_bytes = 2 x FETCH + WRITE (gfx950 tallies a 128-byte read request as 64 bytes: MI355X_MICROARCH.md, HBM).  Launches that returned at a gate
(PCG launches enqueued past convergence: a few KB) are dropped: only dispatches above 5 % of the kernel's largest one count."""
import csv, glob, json, os, statistics, sys

out = sys.argv[1]
want = ("k_tail_sym", "k_tail_sym_fin", "k_tri_wide", "k_tail_mv", "kq_pcg_Gp", "kq_pcg_Aty_lds", "kq_pcg_Aty", "kq_pcg_update", "k_cg_init_A", "k_cg_init_At",
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
json.dump(res, sys.stdout, indent=1)
