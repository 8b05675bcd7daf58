#!/usr/bin/env python
"""bench.py -- ADMM iterations/s of the MI355X-native ABIP-LP hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W [--workload c4|c2|c3|c5] [--no-to-tol] [--no-cpu]

One "step" = one inner ADMM iteration of the real solver trajectory (KKT solve incl. all its PCG
iterations, barrier prox, dual update, averages, stopping test; outer-iteration work -- residuals,
mu update, Barzilai-Borwein search -- runs when the trajectory reaches it and is inside the timed region).
The problem is resident in HBM before the timed region starts (abip_init + abip_hip_solve_begin).

Workloads (BASELINE.json configs; no Netlib/Mittelmann files exist offline, so C2/C3 are seeded
structure-matched surrogates, labelled as such):
    c4  synthetic random sparse LP m=200k n=500k nnz~5M, PCG back-end        (default: the roofline config
        and the only LP config that partitions over GPUs)
    c2  25fv47-class block-staircase LP (816 x 1879, nnz~1e4), direct LDL' back-end
    c3  pds-class multi-commodity network LP, PCG back-end
    c5  LASSO-as-SOCP through the conic path (abip_qcp, p=10000 samples, d=45000 features, n=100002), direct LDL' with a dense tail;
        abip_qcp() is one call (the reference's conic entry point has no stepping form), so a step count cannot be imposed:
        one untimed full solve warms up, a second one is timed and `steps` is the number of ADMM iterations it took

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# stdout carries exactly ONE line, the JSON record: the native libraries (ours mirrors the reference's unconditional "Done the pc
# rescaling!" chatter, the CPU baseline is the reference itself) write to the C-level stdout, which is therefore pointed at stderr
# for the whole run; emit() writes the record to the real stdout at the end.
_REAL_STDOUT = os.dup(1)
os.dup2(2, 1)


def emit(record: dict) -> None:
    sys.stdout.flush()
    os.write(_REAL_STDOUT, (json.dumps(record) + "\n").encode())


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def make_workload(name: str):
    from abip_amd import problems
    if name == "c4":
        A, b, c = problems.lp_random_sparse(m=200_000, n=500_000, per_col=16)
        return A, b, c, "indirect", "synthetic random sparse LP m=200000 n=500000 nnz~5.0e6 (BASELINE configs[3]), PCG"
    if name == "c2":
        A, b, c = problems.lp_staircase()
        return A, b, c, "direct", "synth-25fv47-like block-staircase LP 816x1879 nnz~1.0e4 (BASELINE configs[1] surrogate), direct LDL'"
    if name == "c3":
        A, b, c = problems.lp_multicommodity(nodes=1200, arcs=4400, commodities=10)
        return A, b, c, "indirect", "synth-pds-like multi-commodity network LP (BASELINE configs[2] surrogate), PCG"
    raise SystemExit(f"unknown workload {name}")


def b_spmv(R, C, nnz):
    """Algorithmic bytes of one CSR SpMV: values+indices, row pointers, x read, y read-modify-write."""
    return 12 * nnz + 4 * (R + 1) + 8 * C + 16 * R


def cpu_baseline(A, b, c, linsys, budget_s=20.0):
    """The reference itself (oracle/_ref, kind 'reference') or our C restatement (kind 'port'), one host thread,
    on a bounded prefix of the same trajectory."""
    from oracle import pyoracle as po
    kind = "reference" if po.have_ref() else "port"
    which = "ref" if kind == "reference" else "oracle"
    # probe with a few iterations, then size the sample to the budget
    T = 4
    t0 = time.time()
    r = po.solve(which, A, b, c, linsys=linsys, eps=1e-6, max_admm_iters=T)
    probe = max(r.info["solve_time"] / 1e3, 1e-6)
    per_it = probe / max(r.info["admm_iter"], 1)
    T2 = int(min(max(budget_s / per_it, 8), 20000))
    if T2 > 2 * T and (time.time() - t0) < budget_s:
        r = po.solve(which, A, b, c, linsys=linsys, eps=1e-6, max_admm_iters=T2)
    secs = r.info["solve_time"] / 1e3
    its = r.info["admm_iter"]
    return dict(value=its / secs, unit="ADMM iterations/s", cores=1, kind=kind,
                sample=f"first {its} ADMM iterations of the same LP and settings (max_admm_iters={max(T, T2)}), {secs:.2f} s, single thread, gcc -O2")


def bench_c5(args, rank, world, dist, torch):
    """BASELINE configs[4]: the conic path on LASSO-as-SOCP.  The direct back-end does not shard: N > 1 = N replicas."""
    import numpy as np
    from abip_amd import problems, qcp
    p, d = 10_000, 45_000
    data, K = problems.qcp_lasso_socp(p, d)
    stg = dict(eps=1e-3, linsys_solver=1, verbose=0)     # eps 1e-3: the reference's LASSO protocol (scripts/bench-qcp/test_lasso.m:11)
    sol, info0 = qcp.abip_qcp(data, K, stg)               # warm-up: pages the library in, JIT-free but first-touch allocations
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    sol, info = qcp.abip_qcp(data, K, stg)
    torch.cuda.synchronize()
    elapsed = info["solve_time"]                          # seconds inside abip_qcp between set-up (data resident) and get_solution
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps = int(info["admm_iter"])
    f = info["factor"]
    N, lnnz, T = f["N"], f["lnnz"], f["dense_tail"]
    bytes_solve = 2 * (12 * lnnz + 4 * (N + 1) + 16 * N) + 24 * N + 2 * 20 * N      # SURVEY.md 8(d) B_solve_direct
    avg_ms = f["solve_ms_total"] / max(f["solves_timed"], 1)
    ach = bytes_solve / (avg_ms * 1e-3) / 1e9
    roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None,
                kernel="KKT solve of the conic projection: k_perm_in, k_tri_wide (L21 stream), k_tail_mv x2 (dense inv(L22), inv(L22)'), k_dscale, k_tri_wide, k_perm_out",
                avg_launch_us=avg_ms * 1e3, launches=f["solves_timed"], algorithmic_bytes_per_launch=bytes_solve, lnnz=lnnz, dense_tail=T,
                streamed_bytes_per_solve=8 * T * (T + 1) + 12 * f["head_nnz"], levels=f["levels"])
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        # the conic reference needs MKL headers (unbuildable here) and the scalar oracle's LDL' of the full-size KKT matrix takes
        # hours, so the CPU leg runs the oracle on the same generator at p=1000, d=4500 and says so
        from oracle import pyoracle_qcp as pq
        ds, Ks = problems.qcp_lasso_socp(1000, 4500)
        x, y, s_, oi, _ = pq.solve(ds["A"], ds["b"], ds["c"], Ks, eps=1e-3, eps_p=1e-3, eps_d=1e-3, eps_g=1e-3, eps_inf=1e-3, eps_unb=1e-3, linsys_solver=1)
        cpu = dict(value=oi["admm_iter"] / (oi["solve_time"] / 1e3), unit="ADMM iterations/s", cores=1, kind="port",
                   sample=f"REDUCED instance p=1000, d=4500 (n=10002) of the same generator: {oi['admm_iter']} iterations in {oi['solve_time'] / 1e3:.2f} s "
                          f"(+ {oi['setup_time'] / 1e3:.1f} s set-up), oracle/abip_qcp_oracle.c, single thread, gcc -O2")
    if rank == 0:
        beta = sol["x"][p + 2:p + 2 + d] - sol["x"][p + 2 + d:]
        emit(({
            "metric": "ADMM iterations/s", "value": world * steps / elapsed, "unit": "ADMM iterations/s", "n_gpus": world, "steps": steps, "warmup": int(info0["admm_iter"]),
            "ms_per_step": 1e3 * elapsed / max(steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"LASSO-as-SOCP p={p} d={d} density 0.005 (BASELINE configs[4]): n={p + 2 + 2 * d}, m={p + 1}, K.q=[{p + 2}], K.l={2 * d}; conic path, direct LDL'",
                       "linsys": "direct", "eps": 1e-3,
                       "parallelism": "single GPU" if world == 1 else f"{world} independent replicas (the direct back-end does not shard)"},
            "roofline": roof, "cpu_baseline": cpu,
            "time_to_tol": dict(seconds=info["runtime"], setup_s=info["setup_time"], solve_s=info["solve_time"], status=info["status"], admm_iter=steps,
                                ipm_iter=info["ipm_iter"], res_pri=info["res_pri"], res_dual=info["res_dual"], rel_gap=info["gap"]),
            "extra": {"nnz": int(data["A"].nnz), "nonzero_coefficients": int(np.sum(np.abs(beta) > 1e-6)), "pobj": info["pobj"]},
        }))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c4", choices=["c4", "c2", "c3", "c5"])
    ap.add_argument("--linsys", default=None, choices=["direct", "indirect"], help="override the workload's KKT back-end (c2/c3/c4)")
    ap.add_argument("--to-tol", action="store_true", help="(default on one GPU) also run a full solve to eps=1e-6 and report wall-clock")
    ap.add_argument("--no-to-tol", action="store_true", help="skip the full solve to eps=1e-6 (the second half of BASELINE.json's metric)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="bracket the dominant kernels with hipEvents inside the timed K steps (default: in a second pass of K steps right after)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libabip_hip has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("ABIP_BENCH_FORCE_SHARD") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.workload == "c5":
        return bench_c5(args, rank, world, dist, torch)
    steps = args.steps if args.steps is not None else {"c4": 200, "c2": 2000, "c3": 500}[args.workload]
    warmup = args.warmup if args.warmup is not None else {"c4": 20, "c2": 200, "c3": 50}[args.workload]

    from abip_amd import Solver
    from abip_amd import dist as adist
    A, b, c, linsys, desc = make_workload(args.workload)
    if args.linsys and args.linsys != linsys:
        linsys = args.linsys
        desc += f" [back-end overridden: {linsys}]"
    m, n = A.shape
    nnz = A.nnz
    # (ABIP_BENCH_FORCE_SHARD=1 runs the sharded code path with a single rank: a self-test of the N > 1 plumbing on one GPU)
    sharded = (world > 1 or os.environ.get("ABIP_BENCH_FORCE_SHARD") == "1") and linsys == "indirect" and dist is not None
    shard_note = None
    if sharded:
        # RCCL communicator for the solver; rows of A are split over the ranks inside abip_init.  If the communicator cannot
        # be built on some rank, every rank falls back to an independent replica of the full problem (and the line says so).
        ok = 1
        try:
            adist.init_torch()
        except Exception as e:  # noqa: BLE001
            ok, shard_note = 0, f"sharding unavailable ({e}); ran {world} replicas"
        t = torch.tensor([ok], device="cuda", dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            if ok:
                adist.finalize()
            sharded, shard_note = False, shard_note or f"sharding unavailable on another rank; ran {world} replicas"

    S = Solver(A, b, c, linsys=linsys, eps=1e-6, verbose=0)
    S.begin()
    fin, done_w = S.step(warmup)
    S.sync()
    dom = ("spmv_At", "spmv_A") if linsys == "indirect" else ("sptrsv",)
    if args.events_in_timed_region:
        S.profile_enable(dom)
    S.profile_read(reset=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    fin, done = S.step(steps)
    S.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = S.profile_read(reset=True)
    if not args.events_in_timed_region and not fin:
        # second pass: the same number of steps with hipEvents around every launch of the dominant kernels (two event
        # records per launch cost ~20 % of wall time on this launch-dense path, so they are kept out of `value`)
        S.profile_enable(dom)
        S.step(steps)
        S.sync()
        pe = S.profile_read(reset=True)
        for key in ("ms", "launches"):
            prof[key] = pe[key]
        prof["noop_launches"] = pe["noop_launches"]
        prof["noop_ms"] = pe["noop_ms"]
        events_pass = dict(steps=steps, cg_iters_per_step=pe["cg_iters"] / max(pe["admm_iters"], 1))
    else:
        events_pass = None
    S.profile_enable(())
    if done != steps:
        # the solve terminated inside the timed window: the number is still exact for `done` steps
        steps_eff = done
    else:
        steps_eff = steps
    # N > 1, PCG: ONE problem, rows of A sharded over the ranks (strong scaling: total work fixed).
    # N > 1, direct: the LDL' solve does not shard -> N independent replicas (weak scaling).
    value = steps_eff / elapsed if sharded or world == 1 else world * steps_eff / elapsed

    roof = None
    if linsys == "indirect":
        m_loc = m // world if sharded else m          # this rank's share (rows balanced by non-zeros)
        nnz_loc = nnz // world if sharded else nnz
        cand = {"spmv_At": (b_spmv(n, m_loc, nnz_loc), "k_cg_spmv_At / k_spmv_set (tmp = A'(z + beta p), CSC gather over n rows)"),
                "spmv_A": (b_spmv(m_loc, n, nnz_loc), "k_cg_spmv_A (Gp = A tmp + rho p, CSR gather over m rows)")}
        name = max(cand, key=lambda k: prof["ms"][k])
        nl = max(prof["launches"][name], 1)
        avg_ms = prof["ms"][name] / nl
        ach = cand[name][0] / (avg_ms * 1e-3) / 1e9
        traffic, tsrc = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # PMC counters need rocprofv3: measured in separate passes, committed
        if args.workload == "c4" and world == 1 and os.path.exists(pmc):
            rec = json.load(open(pmc)).get("k_cg_" + name, {})
            traffic, tsrc = rec.get("traffic_bytes"), "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_source=tsrc,
                    kernel=cand[name][1], avg_launch_us=avg_ms * 1e3, launches=nl, algorithmic_bytes_per_launch=cand[name][0])
        # for comparison with a rocprofv3 --stats CSV, whose per-kernel average also counts the launches enqueued past PCG
        # convergence (~3.6 us no-ops): both PCG SpMV kernels, all launches
        n_all = prof["launches"]["spmv_At"] + prof["launches"]["spmv_A"] + prof["noop_launches"]
        if n_all:
            roof["avg_us_both_spmv_incl_noop_launches"] = 1e3 * (prof["ms"]["spmv_At"] + prof["ms"]["spmv_A"] + prof["noop_ms"]) / n_all
            roof["noop_launches"] = prof["noop_launches"]
    else:
        lnnz = int(S.scalar("lnnz")); N = m + n
        bytes_solve = 2 * (12 * lnnz + 4 * (N + 1) + 16 * N) + 24 * N + 2 * 20 * N   # SURVEY.md 8(d) B_solve_direct
        # one solve = the launches of class "sptrsv" between two k_rhs: head levels, the two dense tail mat-vecs, head levels
        nl = max(prof["launches"]["sptrsv"], 1)
        nsolve = max(prof["kkt_solves"], 1)
        avg_ms = prof["ms"]["sptrsv"] / nsolve
        ach = bytes_solve / (avg_ms * 1e-3) / 1e9
        T = int(S.scalar("tail"))
        roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None,
                    kernel="direct solve P' L^-T D^-1 L^-1 P (k_ldl_fwd_small / k_tail_mv x2 / k_ldl_bwd_small, or the segmented level kernels)",
                    avg_launch_us=avg_ms * 1e3, launches=nsolve, kernel_launches=nl, algorithmic_bytes_per_launch=bytes_solve, lnnz=lnnz,
                    dense_tail=T, dense_tail_bytes_per_solve=8 * T * (T + 1), levels=[int(S.scalar("levels_fwd")), int(S.scalar("levels_bwd"))])

    extra = dict(cg_iters_per_step=prof["cg_iters"] / max(prof["admm_iters"], 1), noop_launches=prof["noop_launches"], events_pass=events_pass,
                 m=m, n=n, nnz=int(nnz))
    if linsys == "indirect":
        cg = extra["cg_iters_per_step"]
        l = m + n + 1
        b_cg = b_spmv(n, m, nnz) + b_spmv(m, n, nnz) + 8 * (21 * m + n)
        b_vec = 8 * (37 * l + 8 * m + 19 * n)
        b_iter = cg * b_cg + 4 * b_spmv(m, n, nnz) + b_vec            # SURVEY.md 8(d) "Indirect"
        extra["effective_GBs_whole_iteration"] = b_iter * (steps_eff / elapsed) / 1e9
    S.close()

    tt = None
    # the metric's second half, time to eps = 1e-6 with status Solved: one GPU only (a sharded solve needs every rank in it)
    if (args.to_tol or world == 1) and not args.no_to_tol and rank == 0 and world == 1:
        S2 = Solver(A, b, c, linsys=linsys, eps=1e-6, verbose=0)
        t1 = time.perf_counter()
        info = S2.solve()
        tt = dict(seconds=time.perf_counter() - t1, setup_s=info["setup_time"] / 1e3, solve_s=info["solve_time"] / 1e3,
                  status=info["status"], admm_iter=info["admm_iter"], ipm_iter=info["ipm_iter"],
                  res_pri=info["res_pri"], res_dual=info["res_dual"], rel_gap=info["rel_gap"])
        S2.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(A, b, c, linsys)

    if rank == 0:
        out = {
            "metric": "ADMM iterations/s", "value": value, "unit": "ADMM iterations/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / max(steps_eff, 1), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "linsys": linsys, "eps": 1e-6,
                       "parallelism": "single GPU" if world == 1 else
                       (f"rows of A sharded over {world} GPUs, RCCL all-reduce of A'-partials and packed scalars" if sharded
                        else f"{world} independent replicas (the direct back-end does not shard)")},
            "roofline": roof, "cpu_baseline": cpu, "time_to_tol": tt, "extra": extra,
        }
        if shard_note:
            out["config"]["note"] = shard_note
        emit(out)
    if sharded:
        adist.finalize()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
