/*
 * abip_hip.h -- device-level C ABI of libabip_hip.so (handle-based, host pointers only).
 *
 * abip.h carries the reference's own entry points (abip_init / abip_solve / ...).
 * This header adds what the reference does not have because its loop never leaves
 * one CPU thread: a stepping interface over the SAME solver state machine
 * (src/abip-lp/src/abip.c:2056-2297), unit-level access to the device kernels of
 * the hot path for parity tests, and timing hooks for bench.py.  Callers: the
 * library's own abip_solve(), tests/, bench.py.  No torch / HIP types appear.
 *
 * Reference site each entry point stands for:
 *   abip_hip_solve_begin .. update_work, abip.c:1843-1927 (+ the loop prologue 2084-2100)
 *   abip_hip_step ......... inner/outer loop body, abip.c:2102-2294
 *   abip_hip_solve_end .... get_solution / get_info, abip.c:1296-1414
 *   abip_hip_accum_by_A / _Atrans .. linsys/common.c:598-695 (y += A x, y += A' x)
 *   abip_hip_kkt_solve .... ABIP(solve_lin_sys), linsys/direct.c:305-328, linsys/indirect.c:393-434
 *   abip_hip_get_vector ... read-only view of ABIPWork's vectors, include/abip.h:126-176
 */
#ifndef ABIP_HIP_DEVICE_H
#define ABIP_HIP_DEVICE_H

#include "abip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these declarations ARE its export list */
#endif

/* 0 when a usable gfx950 device is present, else a negative code; fills name (<=127 chars). */
int abip_hip_device_info(char *name, int name_len, long *total_mem_bytes, int *num_cu);

/* --- stepping (abip_solve == begin; while (!step) ; end) ------------------- */
/* Scale b, c, build h, g = K^{-1} h, cold/warm start.  0 on success. */
abip_int abip_hip_solve_begin(ABIPWork *w, const ABIPData *d, const ABIPSolution *sol, ABIPInfo *info);
/* Advance by at most max_admm_steps inner ADMM iterations (outer-iteration work that
 * falls between them -- residuals, mu update, BB search -- is executed as it comes).
 * Returns 0 while the solve is unfinished, 1 once it has terminated (info->status_val
 * then holds the reference's status).  *steps_done (may be NULL) = iterations executed. */
abip_int abip_hip_step(ABIPWork *w, abip_int max_admm_steps, abip_int *steps_done, ABIPInfo *info);
/* Extract (x, y, s) and info exactly as get_solution does; callable at any time. */
abip_int abip_hip_solve_end(ABIPWork *w, ABIPSolution *sol, ABIPInfo *info);

/* --- unit-level kernels (host vectors in, host vectors out) ---------------- */
/* y (m) += A x (n)  /  y (n) += A' x (m), on the SCALED matrix held by w. */
abip_int abip_hip_accum_by_A(ABIPWork *w, const abip_float *x, abip_float *y);
abip_int abip_hip_accum_by_Atrans(ABIPWork *w, const abip_float *x, abip_float *y);
/* rhs (m+n) <- K^{-1} rhs with K = [[rho_y I, A],[A', -I]]; warm (m) may be NULL; iter as in
 * solve_lin_sys (-1 = setup accuracy).  Returns CG iterations used (0 for the direct back-end), <0 on error. */
abip_int abip_hip_kkt_solve(ABIPWork *w, abip_float *rhs, const abip_float *warm, abip_int iter);
/* Copy a device vector to out.  name: "u","v","u_t","h","g","b","c","D","E","Ax" (scaled CSC values),
 * "u_avgcon","v_avgcon".  Layout of the l-vectors is the reference's [y(m) | x(n) | tau].  On a sharded solve the y block and "b" are this
 * rank's rows; "D" always has the m entries of the whole problem (see abip_hip_dist_rows).  Returns length or -1. */
abip_int abip_hip_get_vector(ABIPWork *w, const char *name, abip_float *out, abip_int cap);
/* Scalars: "mu","beta","sigma","gamma","g_th","sc_b","sc_c","nm_b","nm_c","tot_cg_its","lnnz","levels_fwd","levels_bwd","admm_iter","ipm_iter". */
abip_float abip_hip_get_scalar(ABIPWork *w, const char *name);

/* --- multi-GPU (one process per GPU) ---------------------------
 * The PCG back-end shards A by contiguous row blocks (balanced by non-zeros) over `world` ranks: m-space vectors are
 * local, n-space vectors replicated; each rank's A_g' y_g is all-reduced together with the packed reduction scalars.
 * Call ONE of the two init functions on every rank before abip_init; abip_init then takes the FULL problem on every
 * rank (the scaling of A needs all of it) and keeps only its row block on the GPU.  abip_solve returns the full
 * (x, y, s) on every rank.  The direct back-end does not shard: with it every rank is an independent replica.
 *   RCCL:     rank 0 calls abip_hip_dist_get_unique_id, the host program broadcasts the 128 bytes (MPI,
 *             torch.distributed, ...), every rank calls abip_hip_dist_init_rccl after selecting its device.
 *   callback: host-staged sum through a caller-supplied collective (tests; any number of ranks may share one GPU).
 * The solve inside that iteration runs in its COLUMN form by default: the PCG gathers its right-hand side into a replicated m-vector and uses A by column
 * blocks -- one all-reduce of m doubles per PCG iteration instead of n + scalars.  ABIP_HIP_DIST_CG=rows (environment, read by abip_init) keeps the solve on
 * the row blocks too (the all-reduce described above).
 * The conic entry point abip_qcp() (include/abip_qcp.h) uses the same context: with the generic formulation (prob_type 2) and the PCG
 * back-end (linsys_solver 3) it shards the COLUMNS of A over the ranks, cut at cone boundaries (m-space replicated, one all-reduce of m
 * doubles per PCG iteration); every rank passes the full problem and receives the full (x, y, s), bit-identical across the ranks.
 * Any other conic configuration runs as independent replicas.
 * The transport must deliver the SAME bits of every all-reduced vector to every rank (the ranks take their control decisions from their own copies):
 * ring, tree and all-pairs schedules do; abip_hip_dist_init_rccl sets RCCL_MSCCLPP_ENABLE=0 unless the caller has set it, because a one-shot
 * "every rank adds all peers itself" kernel sums in a rank-dependent order.  A callback transport has to honour the same rule. */
typedef void (*abip_hip_allreduce_fn)(void *ctx, double *host_buf, long count); /* in-place sum over all ranks */
int abip_hip_dist_get_unique_id(void *out128);
int abip_hip_dist_init_rccl(int rank, int world, const void *unique_id128);
int abip_hip_dist_init_callback(int rank, int world, abip_hip_allreduce_fn fn, void *ctx);
/* A third transport: a hand-rolled all-reduce over peer-mapped device buffers (abip_amd/csrc/dev_peer.h: one-shot reduce-scatter + all-gather, every chunk
 * summed in ONE place in rank order -- deterministic, the same bits on every rank -- one kernel launch per collective, no library in between; what SURVEY 8(e)
 * prices for xGMI).  Every rank: abip_hip_dist_peer_prepare (allocates its mailbox for vectors of up to cap_doubles -- abip_hip_dist_peer_capacity(m, n) is
 * enough for an LP -- and returns the 64-byte IPC handle); the host program gathers the handles of all ranks in rank order; every rank:
 * abip_hip_dist_init_peer.  At most 8 ranks (one node).  Selected by the host program; RCCL stays the default of bench.py until hardware says otherwise.
 * abip_hip_dist_init_peer returns -4 (and the transport is not set up: use RCCL) when this rank's mailbox could not be allocated as fine-grained memory and a
 * peer sits on another device -- remote writes into coarse-grained memory a local kernel polls have no coherence guarantee; EVERY rank must learn of one rank's
 * refusal before any of them solves (abip_amd/dist.py: init_peer gathers the return codes).  ABIP_HIP_PEER_COARSE_OK=1 overrides. */
long abip_hip_dist_peer_capacity(long m, long n);
int abip_hip_dist_peer_prepare(long cap_doubles, void *handle_out64);
int abip_hip_dist_init_peer(int rank, int world, const void *handles64_by_rank);
void abip_hip_dist_finalize(void);
/* Ranks in the live communicator as the transport reports them (RCCL: ncclCommCount; callback: the world given; none: 0). */
int abip_hip_dist_comm_count(void);
/* Failure protocol of a sharded solve: a rank that hits an error (HIP, a failed collective, an invalid LOQO product) aborts the
 * communicator (ncclCommAbort) before it returns ABIP_FAILED, so that its peers fail instead of waiting in a collective.  The
 * caller must then exit non-zero and let the launcher tear the job down; the library does not retry in-process. */
/* Row ranges the sharded path uses: bounds[g] .. bounds[g+1] are rank g's rows (world+1 entries out).  Pure host code. */
int abip_hip_dist_partition(const ABIPMatrix *A, int world, abip_int *bounds);

/* Pure host code (no device needed; CPU tests of the set-up path): order and factor K = [[rho_y I, A],[A', -I]] as abip_init does
 * for the direct back-end (reference: factorize, linsys/direct.c:218-270) with the dense tail chosen automatically (tail = -1),
 * disabled (0) or forced (T), complete the factorisation on the host and overwrite rhs (m+n) with K^-1 rhs.
 * stats8 = { N, nnz(L), T, forward levels, backward levels, nnz of the sparse head, 0, 0 }.  0 on success. */
int abip_hip_host_factor_solve(const ABIPMatrix *A, double rho_y, int tail, double *rhs, double *stats8);
/* Pure host code: ABIP(_normalize_A) (linsys/common.c:150-565) exactly as abip_init applies it -- A is scaled in place, D (m) and E (n)
 * receive the scaling vectors, means2 the mean row / column norms of the scaled matrix.  0 on success. */
int abip_hip_host_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, double *D, double *E, double *means2);
/* this rank's row range [row0, row1) of the last abip_init (0, m on a single GPU) */
void abip_hip_dist_rows(ABIPWork *w, abip_int *row0, abip_int *row1);

/* Pure host code (runs without a GPU; CPU-side tests of the host logic): the plan of the persistent launch (abip_amd/csrc/dev_xcd.h) for an LP with the
 * sparsity pattern (m, n, Ap, Ai) of A in CSC form and KKT back-end linsys.  out8 = {admitted 0 / 1, workgroups, XCDs, non-zeros per thread, rows of A per
 * thread, rows of A' per thread, LDS bytes per workgroup, rows of the dense inverse kept in LDS}; mb / nb (each workgroups + 1 ints, room for 257; may be
 * NULL) receive the row boundaries of the workgroups' slices of A / A'.  ABIP_HIP_XCD_G in the environment forces the workgroup count as it does in
 * abip_init.  Returns 0, < 0 on invalid arguments. */
int abip_hip_xcd_plan(abip_int m, abip_int n, const abip_int *Ap, const abip_int *Ai, int linsys, double *out8, int *mb, int *nb);
/* Pure host code: the plan of the dense tail's stream (abip_amd/csrc/dev_tail.h) for a tail of T pivots (T % 64 == 0) and `waves` wavefronts wanted: out4 = {column
 * chunks of 512, units of four rows, wavefronts used, slots of the column-partial table}; pre (chunks + 1: units in front of a chunk), qlo / qhi (chunks: the
 * wavefronts whose ranges meet a chunk) may be NULL (room for 65 / 64 / 64 ints).  Returns 0, -1 where no plan exists (T not a multiple of 16, T > 32 768). */
int abip_hip_tail_plan(int T, int waves, int *out4, int *pre, int *qlo, int *qhi);
/* Unit-level access to the direct back-end's LDL' as the LP and the conic path use it: K symmetric quasi-definite, given by its UPPER triangle in CSC
 * form (32-bit indices); rhs (N) <- K^-1 rhs.  on_device 0: everything on the host (no GPU needed); 1: sparse head on the host, dense tail factored and
 * the solve applied on the device.  tail: -1 automatic, 0 none, T forced.  stats4 (may be NULL) = {T, nnz(L), forward levels, backward levels}.
 * Held against the reference's QDLDL (src/external/qdldl/src/qdldl.c: QDLDL_factor / QDLDL_solve, the conic solve of linsys.c:310-316) by tests/test_qdldl_pin*.py. */
int abip_hip_ldl_solve(int N, const int *Kp, const int *Ki, const double *Kx, int tail, int on_device, double *rhs, double *stats4);
/* Unit-level access to the device-side transposition the conic set-up uses for large operators (abip_amd/csrc/dev_transpose.hip): the CSC arrays of an
 * nrows x ncols matrix (32-bit indices) are uploaded, transposed on the device, and the CSR arrays copied back -- out_ptr (nrows + 1), out_col / out_val
 * (Ap[ncols]), the entries of a row in ascending column order, exactly as the host's counting sort (reference: indirect.c:81-139) leaves them.
 * 0 on success, < 0: invalid arguments or no device. */
int abip_hip_csc_to_csr(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, int *out_ptr, int *out_col, double *out_val);

/* --- measurement ------------------------------------------------------------ */
/* Kernel classes timed with hipEvents on the solver's own stream. */
#define ABIP_HIP_K_SPMV_AT 0   /* k_cg_spmv_At: tmp = A'(z + beta p)  (CSC gather, n rows) -- PCG SpMV 1 */
#define ABIP_HIP_K_SPMV_A 1    /* k_cg_spmv_A : Gp = A tmp + rho p    (CSR gather, m rows) -- PCG SpMV 2 */
#define ABIP_HIP_K_CG_VEC 2    /* k_cg_update : x,r,z update + 2 reductions */
#define ABIP_HIP_K_SPTRSV 3    /* permuted L / D / L' solve (direct back-end) */
#define ABIP_HIP_K_VEC 4       /* fused rhs / barrier prox / dual / averages passes, finalize */
#define ABIP_HIP_K_QNORM 5     /* residual SpMV pair + reductions */
#define ABIP_HIP_K_CG_EDGE 6   /* PCG set-up and back-substitution SpMVs (k_cg_init_At/_A, k_post_At) */
#define ABIP_HIP_K_XCD 7       /* k_lp_xcd: the whole inner loop of a cache-resident LP as one persistent launch on one XCD (one "launch" = up to 2048 iterations) */
#define ABIP_HIP_K_CLASSES 8
typedef struct {
  double ms[ABIP_HIP_K_CLASSES];     /* summed device time per class, milliseconds (launches that did work) */
  long launches[ABIP_HIP_K_CLASSES]; /* launches per class that did work */
  double noop_ms;                    /* time of PCG launches enqueued past convergence (they return at their gate) */
  long noop_launches;
  long admm_iters;                   /* inner iterations covered */
  long cg_iters;                     /* CG iterations covered */
  long kkt_solves;                   /* linear solves covered */
  /* device-side stamps (abip_hip_profile_enable_stamps): per class, first-workgroup-begin .. last-workgroup-end of the launches that did work */
  double stamp_ms[ABIP_HIP_K_CLASSES];
  long stamp_launches[ABIP_HIP_K_CLASSES];
  long stamp_noop_launches;          /* stamped launches that returned at a gate (enqueued past PCG convergence) */
  /* sharded solves: the collectives issued by this rank (all of them counted; timed with hipEvents on the solver's stream when bit ABIP_HIP_K_CLASSES of the mask is set, RCCL only) */
  double allreduce_ms;
  long allreduce_calls;
  double allreduce_bytes;
  long cg_iters_skipped;             /* of cg_iters: counted as the reference counts them, NOT executed -- a look-ahead of the Barzilai-Borwein search whose penalty did not
                                      * change hands its second solve to the next look-ahead, whose first solve would repeat it bit for bit (adaptive.c:233-247) */
} AbipHipProfile;
/* mask = bitmask of classes to bracket with events (0 disables).  Timing a class adds two event
 * records per launch of that class only. */
void abip_hip_profile_enable(ABIPWork *w, unsigned mask);
void abip_hip_profile_read(ABIPWork *w, AbipHipProfile *out, int reset);
/* The same durations without event records: the kernels of the masked classes (ABIP_HIP_K_SPMV_AT, ABIP_HIP_K_SPMV_A) write wall-clock
 * ticks from their first and last sampled workgroup into a device ring that rides back with the once-per-iteration control read.
 * Cheap enough to stay on inside a timed region (bench.py does).  0 on success. */
int abip_hip_profile_enable_stamps(ABIPWork *w, unsigned mask);
/* Block until everything queued on the solver's stream has finished. */
void abip_hip_sync(ABIPWork *w);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ABIP_HIP_DEVICE_H */
