#!/usr/bin/env python
"""bench.py -- ADMM iterations/s of the MI355X-native ABIP-LP hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c4|c2|c3|c5|lasso] [--no-to-tol] [--no-cpu] [--no-extra]

One "step" = one inner ADMM iteration of the real solver trajectory (KKT solve incl. all its PCG
iterations, barrier prox, dual update, averages, stopping test; outer-iteration work -- residuals,
mu update, Barzilai-Borwein search -- runs when the trajectory reaches it and is inside the timed region).
The problem is resident in HBM before the timed region starts (abip_init + abip_hip_solve_begin).

Launching.  `--gpus 1` runs in this process.  `--gpus N` (N > 1) under a launcher (torchrun / the driver: RANK, WORLD_SIZE, ... in
the environment) is one rank of N.  `--gpus N` WITHOUT such an environment starts the N ranks itself -- `python -m
torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a child process, before this process has touched the
GPU -- forwards rank 0's JSON line and exits with the child's code.  A node with fewer than N GPUs, a rank that fails or a
WORLD_SIZE that contradicts --gpus is an error (non-zero exit, no JSON line): there is no silent replica fall-back.
ABIP_BENCH_TRANSPORT=gloo-callback puts every rank on cuda:0 with the host-staged collective over gloo (a plumbing dry run
for one-GPU boxes; the line says so and is not a scaling number).

Workloads (BASELINE.json configs; no Netlib/Mittelmann files exist offline, so C2/C3 are seeded
structure-matched surrogates, labelled as such):
    c4  synthetic random sparse LP m=200k n=500k nnz~5M, PCG back-end        (default: the roofline config
        and the only LP config that partitions over GPUs; N > 1 shards its rows: strong scaling)
    c2  25fv47-class block-staircase LP (816 x 1879, nnz~1e4), direct LDL' back-end
    c3  pds-class multi-commodity network LP, PCG back-end
    c5  LASSO-as-SOCP through the conic path (abip_qcp, p=10000 samples, d=45000 features, n=100002), direct LDL' with a dense tail;
        abip_qcp() is one call (the reference's conic entry point has no stepping form), so a step count cannot be imposed:
        one untimed full solve warms up, a second one is timed and `steps` is the number of ADMM iterations it took
    lasso  the reference's own LASSO benchmark (scripts/bench-qcp/test_lasso.m: 5000 x 15000, density 0.15, eps 1e-3) through the LASSO
        front end (abip_ml, prob_type 0; lasso_config.c's formulation and scaling) on the conic device path; y-space PCG unless --linsys direct
On one GPU the default (c4) line also carries short c2 and c3 records under extra.configs (each with its own roofline and cpu_baseline).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import traceback

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


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def die(msg: str, code: int = 2):
    sys.stderr.write("bench.py: " + msg + "\n")
    sys.stderr.flush()
    os._exit(code)


# ---------------------------------------------------------------------------------------------------------
# launching N ranks
# ---------------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> None:
    """--gpus N > 1 outside a launcher: start the ranks as children of this (GPU-free) process and relay rank 0's line."""
    import torch  # device_count() does not initialise the GPU on this image
    transport = os.environ.get("ABIP_BENCH_TRANSPORT", "rccl")
    if transport not in ("rccl", "gloo-callback", "peer"):
        die(f"unknown ABIP_BENCH_TRANSPORT={transport!r} (rccl | gloo-callback | peer)")
    ndev = torch.cuda.device_count()
    if transport == "rccl" and ndev < args.gpus:
        die(f"--gpus {args.gpus} needs {args.gpus} GPUs on this node, found {ndev}: not launching (one process per GPU over RCCL; "
            "ABIP_BENCH_TRANSPORT=gloo-callback runs an N-rank plumbing dry run on one GPU)")
    if ndev < 1:
        die("no GPU visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, timeout=float(os.environ.get("ABIP_BENCH_TIMEOUT", "3000")))
    except subprocess.TimeoutExpired:
        die("the ranks did not finish in time (killed)", 3)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        die(f"the {args.gpus}-rank launch failed (exit code {p.returncode}, {len(lines)} result line(s)); see the ranks' stderr above", p.returncode or 1)
    os.write(_REAL_STDOUT, (lines[-1] + "\n").encode())
    os._exit(0)


# ---------------------------------------------------------------------------------------------------------
# workloads, byte counts, CPU leg
# ---------------------------------------------------------------------------------------------------------
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


def kernel_sources_sha256() -> str:
    """Hash of the sources of the launch path's kernels (abip_amd/_lib.py: dev_kernels.h, every header it includes, solver.hip): ties a committed trace / stamp
    ratio (scripts/trace_medians.py) to the kernels it was measured on."""
    from abip_amd import _lib
    return _lib.kernel_sources_sha256()


def host_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(A, b, c, linsys, budget_s=12.0, window=None):
    """The reference itself (oracle/_ref, kind 'reference') or our C restatement (kind 'port') on ONE host thread, on a bounded prefix
    of the same trajectory; the device is then run under the same iteration cap and its (x, y, s) compared with the CPU leg's (rel_err_xys: the
    parity of THIS workload at THIS size, measured in the same run) -- and the restatement with the reference's OpenMP loop enabled
    (common.c:620-622) on the 4 threads of the reference's own benchmark protocol (scripts/bench-lp/README.md:23-29)."""
    import numpy as np
    from oracle import pyoracle as po
    kind = "reference" if po.have_ref() else "port"
    which = "ref" if kind == "reference" else "oracle"

    def sample(which_, budget):
        T = 4
        t0 = time.time()
        r = po.solve(which_, A, b, c, linsys=linsys, eps=1e-6, max_admm_iters=T)
        per_it = max(r.info["solve_time"] / 1e3, 1e-6) / max(r.info["admm_iter"], 1)
        T2 = int(min(max(budget / per_it, 8), 20000))
        if window:
            T2 = max(T2, min(window, int(4 * budget / per_it)))   # cover the GPU leg's window where that stays bounded
        if T2 > 2 * T and (time.time() - t0) < budget:
            r = po.solve(which_, A, b, c, linsys=linsys, eps=1e-6, max_admm_iters=T2)
        return r, max(T, T2)

    r, cap = sample(which, budget_s)
    its, secs = r.info["admm_iter"], r.info["solve_time"] / 1e3
    rec = dict(value=its / secs, unit="ADMM iterations/s", cores=1, kind=kind, host_cores=host_cores(),
               sample=f"first {its} ADMM iterations of the same LP and settings (max_admm_iters={cap}), {secs:.2f} s, single thread, gcc -O2")
    try:   # the device under the same cap: same counts, same status, how far apart the two (x, y, s) are
        from abip_amd import Solver
        with Solver(A, b, c, linsys=linsys, eps=1e-6, verbose=0, max_admm_iters=cap) as S:
            gi = S.solve()
            rel = lambda a_, b_: float(np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1e-300))
            rec["rel_err_xys"] = max(rel(S.x, r.x), rel(S.y, r.y), rel(S.s, r.s))
            rec["device_same_cap"] = dict(status=gi["status"], ipm_iter=int(gi["ipm_iter"]), admm_iter=int(gi["admm_iter"]), cpu_status=r.info["status"],
                                          cpu_ipm_iter=int(r.info["ipm_iter"]), cpu_admm_iter=int(r.info["admm_iter"]),
                                          counts_equal=bool(gi["ipm_iter"] == r.info["ipm_iter"] and gi["admm_iter"] == r.info["admm_iter"] and gi["status"] == r.info["status"]),
                                          rel_err_pobj=abs(gi["pobj"] - r.info["pobj"]) / (1 + abs(r.info["pobj"])))
    except Exception as e:  # noqa: BLE001
        rec["rel_err_xys"] = None
        rec["device_same_cap"] = dict(error=repr(e))
    if linsys == "indirect":   # the reference's OpenMP site is the SpMV of the PCG path
        try:
            L = po.lib("oracle_omp")
            thr = max(1, min(host_cores(), 4))    # the reference's benchmark protocol runs its solvers on 4 threads (scripts/bench-lp/README.md:23-29)
            L.orc_set_threads(thr)
            r2, cap2 = sample("oracle_omp", budget_s / 2)
            its2, secs2 = r2.info["admm_iter"], r2.info["solve_time"] / 1e3
            slower = " -- SLOWER than one thread on this LP (the loop it parallelises is too short to pay for the fork/join): the single-thread figure stays the baseline" if its2 / secs2 < its / secs else ""
            rec["openmp"] = dict(value=its2 / secs2, unit="ADMM iterations/s", cores=thr, kind="port",
                                 sample=f"first {its2} ADMM iterations, {secs2:.2f} s, oracle/abip_lp_oracle.c built -fopenmp (the reference's own "
                                        f"OpenMP loop, linsys/common.c:620-622; its build must not define _OPENMP), {thr} threads (the reference's bench protocol) of {host_cores()} host cores{slower}")
        except Exception as e:  # noqa: BLE001
            rec["openmp"] = dict(error=str(e))
    return rec


# ---------------------------------------------------------------------------------------------------------
# the conic workload
# ---------------------------------------------------------------------------------------------------------
PMC_FILE = next((f for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f))), "r04_pmc_traffic.json")


def pmc_traffic(case):
    """Per-kernel HBM traffic of the committed counter passes (scripts/r04_pmc.sh -> profiles/r04_pmc_traffic.json; round 3's file before that), or {}."""
    f = os.path.join(ROOT, "profiles", PMC_FILE)
    try:
        return json.load(open(f)).get(case, {})
    except Exception:  # noqa: BLE001
        return {}


def bench_c5(args, rank, world, dist, torch):
    rec = run_conic(args.workload, args.linsys, args.no_cpu, rank, world, dist, torch)
    if rank == 0:
        emit(rec)
    if dist is not None:
        dist.destroy_process_group()


def run_conic(workload, linsys, no_cpu, rank, world, dist, torch):
    """BASELINE configs[4]: the conic path on LASSO-as-SOCP.  The direct back-end does not shard: N > 1 = N replicas."""
    import numpy as np
    from abip_amd import problems, qcp
    class args:  # (the body below was written against the command line)
        pass
    args.workload, args.linsys, args.no_cpu = workload, linsys, no_cpu
    ml = args.workload == "lasso"                         # the reference's own LASSO benchmark through the LASSO front end (abip_ml, prob_type 0)
    if ml:
        p, d = 5000, 15000                                # the largest size of scripts/bench-qcp/test_lasso.m:39-40
        X, yv, lam = problems.lasso_protocol_data(p, d)
        pcg = args.linsys != "direct"                     # PCG unless --linsys direct: the direct back-end spends ~13 s in the host LDL' of this KKT matrix
        stg = dict(prob_type=0, eps=1e-3, linsys_solver=3 if pcg else 1, verbose=0)
        run = lambda: qcp.abip_ml(dict(X=X, y=yv, **{"lambda": lam}), stg)
        nnz_op, m_op, n_op = 1 + p + 2 * int(X.nnz), p + 1, 2 + p + 2 * d
    else:
        p, d = 10_000, 45_000
        data, K = problems.qcp_lasso_socp(p, d)
        pcg = args.linsys == "indirect"                  # --linsys indirect: the conic PCG back-end (linsys_solver 3, abip_amd/csrc/qcp_pcg.h)
        stg = dict(eps=1e-3, linsys_solver=3 if pcg else 1, verbose=0)     # eps 1e-3: the reference's LASSO protocol (scripts/bench-qcp/test_lasso.m:11)
        run = lambda: qcp.abip_qcp(data, K, stg)
        nnz_op = int(data["A"].nnz); m_op, n_op = data["A"].shape
    # N > 1 (or ABIP_BENCH_FORCE_SHARD=1): with the PCG back-end ONE problem is solved, its columns sharded over the ranks (qcp_dist.h: one
    # all-reduce of m doubles per PCG iteration; strong scaling) -- the generic formulation and the LASSO front end alike; the direct back-end
    # runs as N independent replicas
    sharded = dist is not None and pcg
    transport = os.environ.get("ABIP_BENCH_TRANSPORT", "rccl")
    if sharded:
        from abip_amd import dist as adist
        if transport == "rccl":
            adist.init_torch()
        else:
            adist.init_callback(rank, world, lambda arr: dist.all_reduce(torch.from_numpy(arr)))
    sol, info0 = run()                                    # warm-up: pages the library in, JIT-free but first-touch allocations
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    sol, info = run()
    torch.cuda.synchronize()
    if sharded:
        rccl_ranks = adist.comm_count()
        adist.finalize()
    elapsed = info["solve_time"]                          # seconds inside abip_qcp between set-up (data resident) and get_solution
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps = int(info["admm_iter"])
    f = info["factor"]
    N, lnnz, T = f["N"], f["lnnz"], f["dense_tail"]
    bytes_solve = 2 * (12 * lnnz + 4 * (N + 1) + 16 * N) + 24 * N + 2 * 20 * N      # SURVEY.md 8(d) B_solve_direct: what the REFERENCE's algorithm would stream
    avg_ms = f["solve_ms_total"] / max(f["solves_timed"], 1)
    # what THIS back-end streams per solve: the dense tail (one symmetric mat-vec: the lower triangle of inv(S) once + its two partial tables, written and read;
    # or inv(L22) and its transpose), the head's level streams forward and backward (12 bytes per non-zero each), the permuted vectors
    ncc, nwv = (T + 511) // 512, int(os.environ.get("ABIP_HIP_TAIL_WAVES", "2048"))     # dev_tail.h: column chunks, wavefronts of the stream
    tail_bytes = (4 * T * (T + 1) + 2 * 8 * (ncc * T + (nwv + ncc) * 512)) if tail_sym(T) else 8 * T * (T + 1)
    streamed = tail_bytes + 2 * 12 * f["head_nnz"] + 2 * 20 * N
    ach = streamed / (avg_ms * 1e-3) / 1e9
    tr = pmc_traffic("c5_direct") if not ml else {}
    tw = tr.get("k_tri_wide", {}).get("traffic_bytes", 0)
    traffic = (tr["k_tail_sym"]["traffic_bytes"] + tr.get("k_tail_sym_fin", {}).get("traffic_bytes", 0) + tw + tr.get("k_tri_wide_lds", {}).get("traffic_bytes", tw)) if "k_tail_sym" in tr else None
    roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic,
                traffic_source="profiles/" + PMC_FILE + ": k_tail_sym + k_tail_sym_fin + k_tri_wide + k_tri_wide_lds per solve (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" if traffic else None,
                effective_frac=bytes_solve / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                explanation="achieved / frac count the bytes this back-end streams per KKT solve; effective_frac prices the same time against SURVEY 8(d)'s B_solve_direct "
                            "(the two triangular sweeps of the reference's algorithm): the symmetric tail halves the bytes, so the effective figure can exceed the real one",
                kernel="KKT solve of the conic projection: k_perm_in, k_tri_wide (L21 stream), the dense tail (T >= 2048: k_tail_sym + k_tail_sym_fin, the lower triangle of "
                       "inv(S) streamed once, dev_tail.h; else k_tail_mv x2 on inv(L22), inv(L22)'), k_tri_wide_lds (L21' with the tail's solution in LDS; k_tri_wide where it does not fit), k_perm_out",
                avg_launch_us=avg_ms * 1e3, launches=f["solves_timed"], algorithmic_bytes_per_launch=streamed, reference_algorithm_bytes_per_solve=bytes_solve, lnnz=lnnz, dense_tail=T,
                streamed_bytes_per_solve=streamed, levels=f["levels"])
    if pcg:   # one solve = prep + (warm set-up pair) + avg_cg_iters x (A'z, A tn, update) + back-substitution: 2 + 2 + 2 cg + 1 products of the matrix
        nnzA, mA, nA = nnz_op, m_op, n_op
        cg = float(info["avg_cg_iters"])
        bytes_solve = (3 + 2 * cg + 2) * (b_spmv(mA, nA, nnzA) + b_spmv(nA, mA, nnzA)) / 2 + cg * 8 * 8 * mA
        ach = bytes_solve / max(avg_ms * 1e-3, 1e-12) / 1e9
        tr = pmc_traffic("lasso_pcg" if ml else "c5_pcg")
        ka = "kq_pcg_Aty_lds" if "kq_pcg_Aty_lds" in tr else "kq_pcg_Aty"
        traffic = int((cg + 2.5) * (tr[ka]["traffic_bytes"] + tr["kq_pcg_Gp"]["traffic_bytes"]) + cg * tr.get("kq_pcg_update", {}).get("traffic_bytes", 0)) if ka in tr and "kq_pcg_Gp" in tr else None
        roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic,
                    traffic_source=f"profiles/{PMC_FILE}: (cg + 2.5) x ({ka} + kq_pcg_Gp) + cg x kq_pcg_update per solve" if traffic else None,
                    kernel="KKT solve of the conic projection by y-space PCG: kq_pcg_prep, kq_pcg_Aty/_Gp (set-up), {kq_pcg_Aty (m-vector in LDS where it fits: kq_pcg_Aty_lds), kq_pcg_Gp, kq_pcg_update} x cg, kq_pcg_post; one host round trip per solve",
                    avg_launch_us=avg_ms * 1e3, launches=f["solves_timed"], algorithmic_bytes_per_launch=bytes_solve, avg_cg_iters=cg)
    # cpu_baseline: the conic reference does not build here (its sources include MKL headers unconditionally; a stand-in build is not allowed) and the scalar
    # oracle's LDL' of the full-size KKT matrix takes hours, so there is NO CPU leg on this input -- the key stays null (never a different instance
    # under that key).  What the oracle does on a reduced instance of the same generator is kept, labelled as such, under extra.cpu_reduced_instance.
    cpu, cpu_small = None, None
    cpu_why = ("no CPU leg on this input: the conic reference needs MKL headers (unbuildable here, stand-ins not allowed) and the oracle restatement's scalar LDL' "
               "of the full-size KKT matrix takes hours")
    if rank == 0 and world == 1 and not args.no_cpu and ml:
        from oracle import pyoracle_qcp as pq
        Xs, ys, ls_ = problems.lasso_protocol_data(1000, 3000)
        _, oi = pq.solve_lasso(Xs, ys, ls_, eps=1e-3, eps_p=1e-3, eps_d=1e-3, eps_g=1e-3)
        cpu_small = dict(value=oi["admm_iter"] / max(oi["solve_time"] / 1e3, 1e-9), unit="ADMM iterations/s", cores=1, kind="port", host_cores=host_cores(),
                         sample=f"REDUCED instance 1000 x 3000 of the same generator (NOT this record's input): {oi['admm_iter']} iterations in {oi['solve_time'] / 1e3:.2f} s "
                                f"(+ {oi['setup_time'] / 1e3:.1f} s set-up), oracle/abip_qcp_oracle.c (LASSO restatement, dense reduced Cholesky), single thread, gcc -O2")
    elif rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle_qcp as pq
        ds, Ks = problems.qcp_lasso_socp(1000, 4500)
        x, y, s_, oi, _ = pq.solve(ds["A"], ds["b"], ds["c"], Ks, eps=1e-3, eps_p=1e-3, eps_d=1e-3, eps_g=1e-3, eps_inf=1e-3, eps_unb=1e-3, linsys_solver=1)
        cpu_small = dict(value=oi["admm_iter"] / (oi["solve_time"] / 1e3), unit="ADMM iterations/s", cores=1, kind="port", host_cores=host_cores(),
                         sample=f"REDUCED instance p=1000, d=4500 (n=10002) of the same generator (NOT this record's input): {oi['admm_iter']} iterations in {oi['solve_time'] / 1e3:.2f} s "
                                f"(+ {oi['setup_time'] / 1e3:.1f} s set-up), oracle/abip_qcp_oracle.c, single thread, gcc -O2")
    # "eps 1e-3 ... and 1e-6": the same problem once more at the tight tolerance (one GPU)
    tt6 = None
    if rank == 0 and world == 1 and dist is None:
        stg6 = dict(stg, eps=1e-6)
        _, i6 = qcp.abip_ml(dict(X=X, y=yv, **{"lambda": lam}), stg6) if ml else qcp.abip_qcp(data, K, stg6)
        torch.cuda.synchronize()
        tt6 = dict(seconds=i6["runtime"], setup_s=i6["setup_time"], solve_s=i6["solve_time"], status=i6["status"], admm_iter=int(i6["admm_iter"]), ipm_iter=i6["ipm_iter"],
                   res_pri=i6["res_pri"], res_dual=i6["res_dual"], rel_gap=i6["gap"], value=int(i6["admm_iter"]) / max(i6["solve_time"], 1e-12), unit="ADMM iterations/s", eps=1e-6)
    if rank == 0:
        beta = sol["x"] if ml else sol["x"][p + 2:p + 2 + d] - sol["x"][p + 2 + d:]
        wl = (f"LASSO {p} x {d}, density 0.15 (scripts/bench-qcp/test_lasso.m largest size) through the LASSO front end (prob_type 0): conic n={n_op}, m={m_op}, K.rq=[{p + 2}], K.l={2 * d}; "
              if ml else f"LASSO-as-SOCP p={p} d={d} density 0.005 (BASELINE configs[4]): n={p + 2 + 2 * d}, m={p + 1}, K.q=[{p + 2}], K.l={2 * d}; conic path, ")
        return ({
            "metric": "ADMM iterations/s", "value": (steps if sharded else world * steps) / elapsed, "unit": "ADMM iterations/s", "n_gpus": world, "steps": steps, "warmup": int(info0["admm_iter"]),
            "ms_per_step": 1e3 * elapsed / max(steps, 1), "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl + ("y-space PCG" if pcg else "direct LDL'"),
                       "linsys": "indirect (PCG, linsys_solver 3)" if pcg else "direct", "eps": 1e-3,
                       "parallelism": (f"columns of A sharded over {world} ranks at cone boundaries, m-space replicated: one all-reduce of m = {m_op} doubles per PCG iteration, "
                                       f"{int(info['factor']['head_nnz'])} collectives in the solve; transport {transport}" + (f", {rccl_ranks} ranks in the communicator" if transport == "rccl" else " (host-staged: a plumbing dry run, not a scaling number)")) if sharded
                                      else ("single GPU" if world == 1 else f"{world} independent replicas (the direct back-end does not shard)")},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_note": cpu_why, "time_to_tol_eps_1e-6": tt6,
            "time_to_tol": dict(seconds=info["runtime"], setup_s=info["setup_time"], solve_s=info["solve_time"], status=info["status"], admm_iter=steps,
                                ipm_iter=info["ipm_iter"], res_pri=info["res_pri"], res_dual=info["res_dual"], rel_gap=info["gap"]),
            "extra": {"nnz": nnz_op, "nonzero_coefficients": int(np.sum(np.abs(beta) > 1e-6)), "pobj": info["pobj"], "cpu_reduced_instance": cpu_small,
                      "cpu_solve_reference": cpu_solve_reference(avg_ms) if (not ml and not pcg) else None},
        })
    return None


def cpu_solve_reference(device_solve_ms):
    """The kernel-level CPU leg of the conic direct back-end: the reference's own QDLDL_solve on the C5 KKT system, measured ONCE by scripts/qdldl_cpu_leg.py
    (the one-off QDLDL_factor takes minutes) and committed -- a recorded figure with its date and box, NOT a same-run baseline (that key stays null)."""
    f = os.path.join(ROOT, "profiles", "r06_c5_cpu_qdldl_solve.json")
    try:
        r = json.load(open(f))
    except Exception:  # noqa: BLE001
        return None
    return dict(solve_ms=1e3 * r["solve_s_each"], solves_per_s=r["solves_per_s"], factor_s=r["factor_s"], nnz_L=r["nnz_L"], threads=r["threads"], host_cores=r["host_cores"],
                box=r["box"], date=r["date"], source="profiles/r06_c5_cpu_qdldl_solve.json (scripts/qdldl_cpu_leg.py: QDLDL_solve, src/external/qdldl/src/qdldl.c:236-281, oracle/_ref/libqdldl_ref.so)",
                device_solve_ms_this_run=device_solve_ms, ratio=1e3 * r["solve_s_each"] / max(device_solve_ms, 1e-12),
                note="recorded once on the development container's CPU, not on this run's host: never a cpu_baseline")


# ---------------------------------------------------------------------------------------------------------
def tail_sym(T):   # dev_ldl.h: the dense tail applied as one symmetric mat-vec (half the stream of the two triangular ones)
    e = os.environ.get("ABIP_HIP_TAIL_SYM")
    return (int(e) != 0) if e else T >= 2048


# one LP workload: W untimed steps, K timed steps between barriers, roofline of the dominant kernel from device-side stamps
# taken INSIDE the timed window, optionally the full solve to eps 1e-6 and the CPU leg
# ---------------------------------------------------------------------------------------------------------
def run_lp(name, steps, warmup, args, rank, world, dist, torch, sharded, linsys_override=None, to_tol=True, cpu=True, cpu_budget=12.0, pmc=True):
    from abip_amd import Solver
    A, b, c, linsys, desc = make_workload(name)
    if linsys_override and linsys_override != linsys:
        linsys = linsys_override
        desc += f" [back-end overridden: {linsys}]"
    m, n = A.shape
    nnz = A.nnz
    sharded = sharded and linsys == "indirect"

    if dist is not None:
        dist.barrier()
    t_setup = time.perf_counter()
    S = Solver(A, b, c, linsys=linsys, eps=1e-6, verbose=0)      # abip_init: every rank scales the whole A on its host cores, then uploads its share
    t_setup = time.perf_counter() - t_setup
    if dist is not None:
        ts = torch.tensor([t_setup], dtype=torch.float64, device="cuda")
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        t_setup = float(ts.item())
    xcd = S.scalar("xcd") == 1.0       # cache-resident LP: the inner loop runs as the persistent launch on 1, 2 or 4 XCDs (abip_amd/csrc/dev_xcd.h)
    xg = int(S.scalar("xcd_g")) if xcd else 0
    S.begin()
    fin, done_w = S.step(warmup)
    S.sync()
    if sharded:
        S.profile_enable(("allreduce",))                  # two event records per collective
    if xcd:
        S.profile_enable(("xcd",))                        # two event records per launch of up to 2048 iterations: stays on in the timed region
    elif linsys == "indirect":
        S.profile_enable_stamps(("spmv_At", "spmv_A"))    # device-side begin/end ticks, no event records: stays on in the timed region
    elif args.events_in_timed_region:
        S.profile_enable(("sptrsv",))
    S.profile_read(reset=True)
    xstat0 = {k: S.scalar(k) for k in ("xcd_exchanges", "xcd_launches", "xcd_outer_done", "xcd_lookaheads")} if xcd else None

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
    events_pass = None
    if linsys == "direct" and not xcd and not args.events_in_timed_region and not fin:
        # direct back-end: a solve is several launches; bracket them with hipEvents in a second pass of the same length
        # (two event records per launch cost ~20 % of wall time on this launch-dense path, so they are kept out of `value`)
        S.profile_enable(("sptrsv",))
        S.step(steps)
        S.sync()
        pe = S.profile_read(reset=True)
        for key in ("ms", "launches"):
            prof[key] = pe[key]
        prof["kkt_solves_events"] = pe["kkt_solves"]
        events_pass = dict(steps=steps)
    S.profile_enable(())
    steps_eff = done if done != steps else steps   # the solve may terminate inside the window: the number is exact for `done` steps
    # N > 1, PCG: ONE problem, rows of A sharded over the ranks (strong scaling: total work fixed).
    # N > 1, direct: the LDL' solve does not shard -> N independent replicas (weak scaling).
    value = steps_eff / elapsed if sharded or world == 1 else world * steps_eff / elapsed
    row0, row1 = S.rows()

    roof = None
    l = m + n + 1
    cg_step = prof["cg_iters"] / max(prof["admm_iters"], 1)      # as the reference counts them (a look-ahead solve the device does not repeat is booked with its count)
    cg_exec = (prof["cg_iters"] - prof.get("cg_iters_skipped", 0)) / max(prof["admm_iters"], 1)   # what the device executed
    if xcd:
        # SURVEY.md 8(d), per inner iteration: PCG  cg (2 B_spmv + vectors) + 4 B_spmv + vectors;  direct  B_solve of this back-end = the dense
        # inverse of rho I + A A' (8 m^2) + the two products around it, + 2 B_spmv of the stopping test + vectors
        b_cg = b_spmv(n, m, nnz) + b_spmv(m, n, nnz) + 8 * (21 * m + n)
        b_vec = 8 * (37 * l + 8 * m + 19 * n)
        b_iter = (cg_exec * b_cg + 4 * b_spmv(m, n, nnz) + b_vec) if linsys == "indirect" else (8 * m * m + 4 * b_spmv(m, n, nnz) + b_vec)   # executed PCG iterations only
        nl = max(prof["launches"]["xcd"], 1)
        its = max(prof["admm_iters"], 1)
        ms = prof["ms"]["xcd"]
        ach = b_iter * its / max(ms * 1e-3, 1e-12) / 1e9
        # exchanges counted by the kernel itself (every rendez-vous of the window's launches, the look-ahead solves of the Barzilai-Borwein search included:
        # since round 4 the search runs inside the launch, so its time is inside `ms` as well)
        xd = {k: S.scalar(k) - xstat0[k] for k in xstat0}
        exch = xd["xcd_exchanges"] / its if xd["xcd_exchanges"] > 0 else ((2 * cg_step + 6) if linsys == "indirect" else 3.0)
        # HBM bytes per launch from the committed counter passes (per inner iteration there, FETCH_SIZE / WRITE_SIZE summed over the kernel's dispatches)
        tpi = pmc_traffic(name).get("k_lp_xcd", {}).get("traffic_bytes_per_iteration") if pmc and world == 1 else None
        roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=(tpi * its / nl) if tpi else None,
                    traffic_source="profiles/" + PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; per inner iteration x iterations per launch)" if tpi else None,
                    kernel="k_lp_xcd: the whole inner ADMM loop as one persistent launch, one workgroup per CU on %d XCD(s) (slices of A and A' per workgroup; operands "
                           "handed over through the L2%s: stores, acknowledged, partial-sum granules as flags, L1-bypassing gathers); bound by the latency "
                           "of its exchanges -- a chain of L2 round trips and two workgroup barriers -- not by HBM" % (xg // 32, " (written through between XCDs)" if xg > 32 else ""),
                    workgroups=xg, outer_iterations_inside_the_launches=int(xd["xcd_outer_done"]), lookahead_steps_inside_the_launches=int(xd["xcd_lookaheads"]),
                    avg_launch_us=1e3 * ms / nl, launches=nl, iterations_per_launch=its / nl, algorithmic_bytes_per_launch=b_iter * its / nl,
                    algorithmic_bytes_per_iteration=b_iter, us_per_iteration=1e3 * ms / its, exchanges_per_iteration=exch, us_per_exchange=1e3 * ms / its / exch,
                    timing="hipEvents around every launch inside the timed region")
    elif linsys == "indirect":
        m_loc = row1 - row0
        nnz_loc = nnz // world if sharded else nnz
        cand = {"spmv_At": (b_spmv(n, m_loc, nnz_loc), "k_cg_spmv_At / k_spmv_set_t (tmp = A'(z + beta p), CSC gather over n rows)"),
                "spmv_A": (b_spmv(m_loc, n, nnz_loc), "k_cg_spmv_A (Gp = A tmp + rho p, CSR gather over m rows)")}
        kname = max(cand, key=lambda k: prof["stamp_ms"][k])
        nl = max(prof["stamp_launches"][kname], 1)
        avg_ms = prof["stamp_ms"][kname] / nl
        ach = cand[kname][0] / max(avg_ms * 1e-3, 1e-12) / 1e9
        traffic, tsrc = None, None
        if pmc and name == "c4" and world == 1:   # PMC counters need rocprofv3: measured in separate passes (scripts/r03_pmc.sh), committed
            rec = pmc_traffic("c4").get("k_cg_" + kname, {})
            traffic, tsrc = rec.get("traffic_bytes"), "profiles/" + PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        # The driver-parsed figures (achieved, frac, avg_launch_us) are the ones `profiles/` reproduces: kernel-trace durations (dispatch to drain).  The
        # device-side stamps of this run leave out dispatch and drain; they are scaled by the ratio trace / stamp that ONE profiled run measured for the
        # same kernel (scripts/r04_trace.sh -> profiles/r04_trace_durations.json) and kept, unscaled, under `stamps`.
        ratio, tsrc2, tr, stale = 1.0, None, None, None
        for trf in ("r06_trace_durations.json", "r05_trace_durations.json", "r04w_trace_durations.json"):
            path = os.path.join(ROOT, "profiles", trf)
            if name == "c4" and world == 1 and os.path.exists(path):
                blob = json.load(open(path))
                cand_tr = blob.get("k_cg_" + kname)
                if not cand_tr:
                    continue
                if blob.get("kernel_sources_sha256") != kernel_sources_sha256():   # taken on other kernels than the ones built now: not applied
                    stale = f"profiles/{trf} was measured on other kernel sources (sha256 {str(blob.get('kernel_sources_sha256'))[:12]}..., now {kernel_sources_sha256()[:12]}...): ratio not applied, the figures are this run's device-side stamps"
                    continue
                tr = cand_tr
                ratio, tsrc2 = tr["mean_working_us"] / tr["stamp_avg_us_same_run"], "profiles/" + trf
                break
        stamps = dict(avg_launch_us=avg_ms * 1e3, achieved=ach, frac=ach / HBM_PEAK_GBS,
                      timing="device wall-clock ticks (first sampled workgroup begin .. last sampled workgroup end) of every launch that did work, inside the timed region")
        roof = dict(bound="hbm", achieved=ach / ratio, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS / ratio, traffic=traffic, traffic_source=tsrc,
                    kernel=cand[kname][1], avg_launch_us=avg_ms * 1e3 * ratio, launches=nl, algorithmic_bytes_per_launch=cand[kname][0],
                    timing=("kernel-trace equivalent: this run's device-side stamps x (trace duration / stamp duration) of one profiled run of the same kernel" if tr else
                            "device wall-clock ticks (first sampled workgroup begin .. last sampled workgroup end) of every launch that did work, inside the timed region"),
                    trace=(dict(trace_over_stamp=ratio, profiled_run_trace_us=tr["mean_working_us"], profiled_run_stamp_us=tr["stamp_avg_us_same_run"], source=tsrc2) if tr else None),
                    stamps=stamps, trace_file_stale=stale,
                    other_spmv={k: dict(avg_launch_us=1e3 * prof["stamp_ms"][k] / max(prof["stamp_launches"][k], 1), launches=prof["stamp_launches"][k],
                                        algorithmic_bytes_per_launch=cand[k][0], timing="device-side stamps") for k in cand if k != kname},
                    noop_launches=prof["stamp_noop_launches"])
    else:
        lnnz = int(S.scalar("lnnz")); N = m + n
        bytes_solve = 2 * (12 * lnnz + 4 * (N + 1) + 16 * N) + 24 * N + 2 * 20 * N   # SURVEY.md 8(d) B_solve_direct
        # one solve = the launches of class "sptrsv" between two k_rhs: head levels, the two dense tail mat-vecs, head levels
        nl = max(prof["launches"]["sptrsv"], 1)
        nsolve = max(prof.get("kkt_solves_events", prof["kkt_solves"]), 1)
        avg_ms = prof["ms"]["sptrsv"] / nsolve
        ach = bytes_solve / max(avg_ms * 1e-3, 1e-12) / 1e9
        T = int(S.scalar("tail"))
        roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None,
                    kernel="direct solve P' L^-T D^-1 L^-1 P (k_ldl_small<fwd> / k_tail_mv x2 or k_tail_sym / k_ldl_small<bwd>, or the segmented level kernels)",
                    avg_launch_us=avg_ms * 1e3, launches=nsolve, kernel_launches=nl, algorithmic_bytes_per_launch=bytes_solve, lnnz=lnnz,
                    dense_tail=T, dense_tail_bytes_per_solve=8 * T * (T + 1), levels=[int(S.scalar("levels_fwd")), int(S.scalar("levels_bwd"))],
                    timing="hipEvents around the solve's launches in a second pass of the same length" if events_pass else "hipEvents in the timed region")

    coll = None
    if sharded:
        its_ = max(prof["admm_iters"], 1)
        coll = dict(collectives_per_step=prof["allreduce_calls"] / its_, bytes_per_step=prof["allreduce_bytes"] / its_,
                    ms_in_allreduce_per_step=prof["allreduce_ms"] / its_, share_of_step=prof["allreduce_ms"] / its_ / max(1e3 * elapsed / max(steps_eff, 1), 1e-12),
                    timing="hipEvents around every ncclAllReduce on the solver's stream (rank 0's view)")
    extra = dict(collectives=coll, cg_iters_per_step=cg_step, cg_iters_executed_per_step=cg_exec, events_pass=events_pass, m=m, n=n, nnz=int(nnz), rows=[int(row0), int(row1)], persistent_launch=bool(xcd), persistent_launch_workgroups=xg,
                 setup_wall_s=t_setup)   # abip_init, max over the ranks (N ranks on one node run N host passes over A side by side)
    if linsys == "indirect":
        cg = extra["cg_iters_per_step"]
        b_cg = b_spmv(n, m, nnz) + b_spmv(m, n, nnz) + 8 * (21 * m + n)
        b_vec = 8 * (37 * l + 8 * m + 19 * n)
        b_iter = cg * b_cg + 4 * b_spmv(m, n, nnz) + b_vec            # SURVEY.md 8(d) "Indirect"
        extra["effective_GBs_whole_iteration"] = b_iter * (steps_eff / elapsed) / 1e9      # prices the reference-equivalent PCG count (bytes of skipped solves never moved)
        extra["effective_GBs_executed"] = (extra["cg_iters_executed_per_step"] * b_cg + 4 * b_spmv(m, n, nnz) + b_vec) * (steps_eff / elapsed) / 1e9
        extra["cg_iters_note"] = ("cg_iters_per_step counts PCG iterations as the reference does (it repeats a look-ahead solve whose penalty did not change, adaptive.c:233-247); "
                                  "cg_iters_executed_per_step leaves out the solves the device took over bit for bit instead of repeating them")
    S.close()

    tt = None
    # the metric's second half, time to eps = 1e-6 with status Solved: one GPU only (a sharded solve needs every rank in it)
    if to_tol and rank == 0 and world == 1:
        S2 = Solver(A, b, c, linsys=linsys, eps=1e-6, verbose=0)
        t1 = time.perf_counter()
        info = S2.solve()
        tt = dict(seconds=time.perf_counter() - t1, setup_s=info["setup_time"] / 1e3, solve_s=info["solve_time"] / 1e3,
                  status=info["status"], admm_iter=info["admm_iter"], ipm_iter=info["ipm_iter"],
                  res_pri=info["res_pri"], res_dual=info["res_dual"], rel_gap=info["rel_gap"],
                  cg_iters_per_step=S2.scalar("tot_cg_its") / max(info["admm_iter"], 1) if linsys == "indirect" else None,
                  cg_iters_executed_per_step=(S2.scalar("tot_cg_its") - S2.scalar("tot_cg_skipped")) / max(info["admm_iter"], 1) if linsys == "indirect" else None)
        if S2.scalar("xcd") == 1.0:   # how many launches the whole solve took, and how much of the outer loop ran inside them (dev_xcd.h XcdOuter)
            tt.update(launches=int(S2.scalar("xcd_launches")), outer_iterations_inside_the_launches=int(S2.scalar("xcd_outer_done")),
                      lookahead_steps_inside_the_launches=int(S2.scalar("xcd_lookaheads")), launches_abandoned=int(S2.scalar("xcd_giveups")))
        S2.close()

    cpu_rec = None
    if cpu and rank == 0 and world == 1:
        cpu_rec = cpu_baseline(A, b, c, linsys, budget_s=cpu_budget, window=warmup + steps)

    rec = {
        "metric": "ADMM iterations/s", "value": value, "unit": "ADMM iterations/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / max(steps_eff, 1), "higher_is_better": True,
        "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "linsys": linsys, "eps": 1e-6,
                   "parallelism": "single GPU" if world == 1 else
                   (f"rows of A sharded over {world} ranks; PCG form {os.environ.get('ABIP_HIP_DIST_CG', 'default')}: one all-reduce per PCG iteration (rows: the n A'-partials + packed scalars; cols: the m-vector A A'p)" if sharded
                    else f"{world} independent replicas (the direct back-end does not shard)")},
        "roofline": roof, "cpu_baseline": cpu_rec, "time_to_tol": tt, "extra": extra,
    }
    if tt and tt["status"] == "Solved":
        # the whole solve from the start point to eps 1e-6 (every outer iteration, BB search and residual check included): the window-free rate
        rec["value_whole_solve"] = tt["admm_iter"] / tt["solve_s"]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c4", choices=["c4", "c2", "c3", "c5", "lasso"])
    ap.add_argument("--linsys", default=None, choices=["direct", "indirect"], help="override the workload's KKT back-end (c2/c3/c4)")
    ap.add_argument("--to-tol", action="store_true", help="(default on one GPU) also run a full solve to eps=1e-6 and report wall-clock")
    ap.add_argument("--no-to-tol", action="store_true", help="skip the full solve to eps=1e-6 (the second half of BASELINE.json's metric)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the short c2 / c3 records the default one-GPU line carries under extra.configs")
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="direct back-end: bracket the solve's launches with hipEvents inside the timed K steps (default: in a second pass of K steps)")
    args = ap.parse_args()
    if args.gpus < 1:
        die("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)          # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        die(f"--gpus {args.gpus} contradicts WORLD_SIZE={world} (launch one rank per GPU, or drop the launcher and let bench.py start them)")
    transport = os.environ.get("ABIP_BENCH_TRANSPORT", "rccl")
    force_shard = os.environ.get("ABIP_BENCH_FORCE_SHARD") == "1"   # the sharded code path with a single rank: self-test of the N > 1 plumbing
    import torch
    if not torch.cuda.is_available():
        die("needs a GPU (libabip_hip has no CPU path)")
    if transport == "rccl" and world > torch.cuda.device_count():
        die(f"{world} ranks but {torch.cuda.device_count()} GPUs visible")
    # rccl: one GPU per rank.  peer (the hand-rolled exchange over IPC-mapped mailboxes, dev_peer.h): one GPU per rank where the node has them, else every rank on
    # cuda:0 (a functional dry run).  gloo-callback: every rank on cuda:0, host-staged sums.
    peer_one_gpu = transport == "peer" and torch.cuda.device_count() < world
    if peer_one_gpu:
        os.environ.setdefault("ABIP_HIP_PEER_WAIT_MS", "60000")   # the ranks take turns on the one device: a wait inside a kernel may see its peer off it for a while
    torch.cuda.set_device(local_rank if (transport == "rccl" or (transport == "peer" and not peer_one_gpu)) else 0)
    dist = None
    if world > 1 or force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if transport == "rccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    try:
        if args.workload in ("c5", "lasso"):
            return bench_c5(args, rank, world, dist, torch)
        steps = args.steps if args.steps is not None else {"c4": 200, "c2": 2000, "c3": 500}[args.workload]
        warmup = args.warmup if args.warmup is not None else {"c4": 20, "c2": 200, "c3": 50}[args.workload]
        from abip_amd import dist as adist
        linsys0 = args.linsys or {"c4": "indirect", "c3": "indirect", "c2": "direct"}[args.workload]
        sharded = dist is not None and linsys0 == "indirect"
        if sharded:   # communicator for the solver; rows of A are split over the ranks inside abip_init.  Failure here is fatal (no replica fall-back).
            if transport == "rccl":
                adist.init_torch()
            elif transport == "peer":
                shape = {"c4": (200_000, 500_000), "c3": (16_390, 48_400), "c2": (816, 1_879)}[args.workload]
                adist.init_peer_torch(*shape)
            else:
                adist.init_callback(rank, world, lambda arr: dist.all_reduce(torch.from_numpy(arr)))
        if sharded and "ABIP_HIP_DIST_CG" not in os.environ:
            # the headline of an N-GPU line is the library's default form -- the one that exchanges less: columns (one all-reduce of m doubles per PCG iteration, the
            # iteration around the solve on row blocks) whenever m < n; north_star's literal form (rows: all-reduce of the n A'-partials) is measured beside it
            shape = {"c4": (200_000, 500_000), "c3": (16_390, 48_400), "c2": (816, 1_879)}[args.workload]
            os.environ["ABIP_HIP_DIST_CG"] = "cols" if shape[0] < shape[1] else "rows"
        rec = run_lp(args.workload, steps, warmup, args, rank, world, dist, torch, sharded, linsys_override=args.linsys,
                     to_tol=(args.to_tol or world == 1) and not args.no_to_tol, cpu=not args.no_cpu)
        if sharded and os.environ.get("ABIP_BENCH_ONE_FORM") != "1":
            # ... and the column form of the sharded solve, same window, same invocation (the one multi-GPU run the driver may do should show both)
            form0 = os.environ["ABIP_HIP_DIST_CG"]
            os.environ["ABIP_HIP_DIST_CG"] = "cols" if form0 == "rows" else "rows"
            r2 = run_lp(args.workload, steps, warmup, args, rank, world, dist, torch, sharded, linsys_override=args.linsys, to_tol=False, cpu=False)
            rec["extra"]["dist_" + os.environ["ABIP_HIP_DIST_CG"]] = {k: r2[k] for k in ("value", "ms_per_step", "roofline") if k in r2} | dict(collectives=r2["extra"]["collectives"],
                                                                                                                      cg_iters_per_step=r2["extra"]["cg_iters_per_step"], cg_iters_executed_per_step=r2["extra"]["cg_iters_executed_per_step"])
            os.environ["ABIP_HIP_DIST_CG"] = form0
        if dist is not None:
            rows = [None] * world
            dist.all_gather_object(rows, rec["extra"]["rows"])
            rec["rank_rows"] = rows
            rec["transport"] = ({"rccl": "rccl", "peer": "peer-mapped mailboxes (dev_peer.h: one-shot reduce-scatter + all-gather, every chunk summed in one place in rank order)"
                                         + (" -- every rank on cuda:0: a functional dry run, NOT a scaling number" if peer_one_gpu else "")}.get(
                                             transport, "gloo-callback (host-staged sums, every rank on cuda:0: a plumbing dry run, NOT a scaling number)")) if sharded else "none (replicas)"
            rec["rccl_ranks"] = adist.comm_count() if sharded else 0
            # form of the sharded solve: "cols" (default: the solve's m-space gathered and replicated, A by column blocks, one all-reduce of m doubles per PCG
            # iteration) or "rows" (ABIP_HIP_DIST_CG=rows: one all-reduce of n doubles + packed scalars per PCG iteration)
            rec["dist_cg"] = ("cols" if os.environ.get("ABIP_HIP_DIST_CG") == "cols" else "rows") if sharded else None
        if rank == 0 and world == 1 and args.workload == "c4" and not args.no_extra and not force_shard:
            # the Netlib-class and the pds-class configs, short windows, inside the same driver-run line
            sub = {}
            for nm, k_, w_ in (("c2", 2000, 200), ("c3", 300, 50)):
                r = run_lp(nm, k_, w_, args, 0, 1, None, torch, False, to_tol=not args.no_to_tol, cpu=not args.no_cpu, cpu_budget=4.0)
                sub[nm] = {k: r[k] for k in ("value", "value_whole_solve", "ms_per_step", "steps", "warmup", "config", "roofline", "cpu_baseline", "time_to_tol", "extra") if k in r}
            # BASELINE configs[4] (both conic back-ends) and the reference's own LASSO protocol: whole solves at eps 1e-3, seconds each
            for nm, wl_, ls_ in (("c5_direct", "c5", "direct"), ("c5_pcg", "c5", "indirect"), ("lasso", "lasso", None)):
                try:
                    r = run_conic(wl_, ls_, args.no_cpu, 0, 1, None, torch)
                    sub[nm] = {k: r[k] for k in ("value", "ms_per_step", "steps", "config", "roofline", "cpu_baseline", "cpu_baseline_note", "time_to_tol", "time_to_tol_eps_1e-6", "extra")}
                except Exception as e:  # noqa: BLE001 -- a conic record must not cost the line its LP numbers
                    sub[nm] = dict(error=repr(e))
            rec["extra"]["configs"] = sub
        if rank == 0:
            emit(rec)
        if sharded:
            adist.finalize()
        if dist is not None:
            dist.destroy_process_group()
    except BaseException:  # noqa: BLE001 -- a failing rank must take the job down: its peers may be waiting in a collective
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)


if __name__ == "__main__":
    main()
