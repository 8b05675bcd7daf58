/*
 * abip_qcp.h -- C ABI of the conic (ABIP-QCP) path of libabip_hip.so.
 *
 *   min 1/2 x'Qx + c'x   s.t.  Ax = b,  x in K = SOC^q x RSOC^rq x R^f x {0}^z x R^l_+
 *
 * Layout-compatible with the reference's QCP structs (src/abip-qcp/include/abip.h:63-158; the QCP sources are
 * built WITHOUT -DDLONG, so abip_int is a 32-bit int there -- the mex casts mwIndex arrays, abip_qcp_mex.c:186-188)
 * and entry point `abip(d, sol, info, K)` (src/abip-qcp/include/abip.h:235-241, source/abip.c:1335-1371).
 * Because that name and the LP structs' names collide with include/abip.h in one translation unit, this header
 * uses a qcp_ prefix for the types and exports the entry point as abip_qcp(); INTEGRATION.md shows the one-line
 * `#define` a maintainer adds to compile the reference's abip_qcp_mex.c against it.
 *
 * Served formulations (settings.prob_type as abip() maps it, source/abip.c:1341-1348):
 *   2  generic QCP (qcp_config.c): data = (A, Q, b, c), cone K from the caller; sol = (x, y, s).
 *   0  LASSO (lasso_config.c):  min 1/2 |X beta - y|^2 + lambda |beta|_1.   d->A = X (m x n), d->b = y, d->lambda, d->c ignored;
 *      K = {rq: [m + 2], l: 2 n} (as abip_ml_mex.c:328-331 builds it).  sol->x = beta (n entries); sol->y, sol->s are not touched
 *      (the reference frees them and leaves the pointers dangling, lasso_config.c:296-311).
 *   1  soft-margin SVM as an SOCP (svm_config.c):  min 1/2 |w|^2 + lambda sum xi,  y_i (x_i'w + b) >= 1 - xi_i.  d->A = X, d->b = labels,
 *      K = {rq: [n + 2], l: 2 + 2 m + 2 n}.  sol->x = w (n), sol->y = b (1), sol->s = xi (m)  (un_scaling_svm_sol, svm_config.c:410-440).
 *   3  the same SVM as a QP (svm_qp_config.c) with weight 1 / (m lambda) on sum xi; K = {f: n + 1, l: 2 m}; same outputs.
 *   For 0, 1 and 3 the library builds the conic problem (formulation, the formulation's own scaling rule, residual definitions, stopping
 *   test, un-scaling) as the reference's spe_problem vtable does (include/abip.h:27-60), materialises its operator as one sparse matrix and
 *   runs the same device path; the caller's X is never modified (the reference folds the SVM labels into it in place).  They need
 *   normalize = 1 (0 and 1 also scale_E = 1): the reference scales unconditionally and un-scales only under `normalize`.
 * Several GPUs (one process per GPU, include/abip_hip.h "multi-GPU"): prob_type 2 with linsys_solver 3 shards the columns of A at cone
 * boundaries; sol->x / y / s come back whole on every rank (a finite settings.time_limit costs one more small collective per iteration: the ranks vote).
 * Back-ends: a direct solver (sparse LDL' with a dense tail on the device: linsys_solver 1, the reference's QDLDL; the reference's other exact factorisations
 * of the same KKT system -- 0 MKL-DSS, 2 CSparse Cholesky, 4 PARDISO, 5 LAPACK dense Cholesky, which its default rule picks for dense data -- select it too) and a
 * device PCG (linsys_solver 3).  The reference's own PCG for this formulation is unreachable through abip() and ill-posed
 * (abip_amd/csrc/qcp_pcg.h), so linsys_solver 3 is defined here: Jacobi-PCG on the y-space Schur
 * complement rho_y I + A (rho_x I + Q)^-1 A' (the reference's `pcg` of linsys.c:629-716 with H^-1 in the middle; Q absent or
 * diagonal), warm start and tolerance as the reference's projection prepares them (abip.c:206-218).  Its MKL / LAPACKE /
 * CSparse-Cholesky back-ends are not re-implemented: asking for one runs the device LDL' (same solve, same iterates to rounding); values outside 0..5 are rejected with ABIP_FAILED.
 */
#ifndef ABIP_HIP_QCP_H
#define ABIP_HIP_QCP_H

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these declarations ARE its export list */
#endif

typedef int qcp_int;
typedef double qcp_float;

typedef struct { /* struct ABIP_A_DATA_MATRIX, src/abip-qcp/include/amatrix.h */
  qcp_float *x;
  qcp_int *i;
  qcp_int *p;
  qcp_int m, n;
} QCPMatrix;

typedef struct { /* struct ABIP_CONE, abip.h:67-76; columns of A must come in this order */
  qcp_int *q;     /* second-order cones (t, x): t >= ||x||        */
  qcp_int qsize;
  qcp_int *rq;    /* rotated cones (t1, t2, x): 2 t1 t2 >= ||x||^2 */
  qcp_int rqsize;
  qcp_int f;      /* free     */
  qcp_int z;      /* zero     */
  qcp_int l;      /* x >= 0   */
} QCPCone;

typedef struct { /* struct ABIP_SETTINGS, abip.h:93-131 */
  qcp_int normalize;
  qcp_int scale_E;
  qcp_int scale_bc;
  qcp_float scale;
  qcp_float rho_x;
  qcp_float rho_y;
  qcp_float rho_tau;

  qcp_int max_ipm_iters;
  qcp_int max_admm_iters;
  qcp_float eps;
  qcp_float eps_p;
  qcp_float eps_d;
  qcp_float eps_g;
  qcp_float eps_inf;
  qcp_float eps_unb;

  qcp_float err_dif;
  qcp_float alpha;
  qcp_float cg_rate;

  qcp_int use_indirect;
  qcp_int inner_check_period;
  qcp_int outer_check_period;

  qcp_int verbose;
  qcp_int linsys_solver; /* 3 = PCG on the y-space Schur complement (Q absent or diagonal; abip_amd/csrc/qcp_pcg.h); 0, 1, 2, 4, 5 = direct (the device LDL') */
  qcp_int prob_type;     /* 0 LASSO, 1 SVM-SOCP, 2 generic QCP (what abip_qcp_mex.c:436 sets), 3 SVM-QP -- see the header comment */
  qcp_float time_limit;  /* seconds */
  qcp_float psi;

  qcp_int origin_scaling;
  qcp_int ruiz_scaling;
  qcp_int pc_scaling;
} QCPSettings;

typedef struct { /* struct ABIP_PROBLEM_DATA, abip.h:79-91 */
  qcp_int m;
  qcp_int n;
  QCPMatrix *A;
  QCPMatrix *Q; /* full symmetric storage, may be NULL */
  qcp_float *b;
  qcp_float *c;
  qcp_float lambda;
  QCPSettings *stgs;
} QCPData;

typedef struct { qcp_float *x, *y, *s; } QCPSolution; /* abip.h:133-138 */

typedef struct { /* struct ABIP_INFO, abip.h:140-158 */
  char status[32];
  qcp_int status_val;
  qcp_int ipm_iter;
  qcp_int admm_iter;
  qcp_float pobj;
  qcp_float dobj;
  qcp_float res_pri;
  qcp_float res_dual;
  qcp_float rel_gap;
  qcp_float res_infeas;
  qcp_float res_unbdd;
  qcp_float setup_time; /* ms */
  qcp_float solve_time; /* ms */
  qcp_float avg_linsys_time;
  qcp_float avg_cg_iters;
} QCPInfo;

/* abip(d, sol, info, K) of the reference (source/abip.c:1335-1371).  sol->x/y/s are malloc'ed when NULL (sizes per formulation above). */
qcp_int abip_qcp(const QCPData *d, QCPSolution *sol, QCPInfo *info, QCPCone *K);
/* ABIP(set_default_settings), source/util.c:203-255 (prob_type is left at the mex's value 2 = QCP). */
void abip_qcp_set_default_settings(QCPData *d);
/* Shape of the factor and KKT-solve time of the last abip_qcp() call (bench, profiling; no reference counterpart):
 * out8 = { N, dense-tail size T, nnz(L), forward levels, backward levels, solves timed, total ms of those solves (hipEvents
 * on the solver's stream), nnz of the sparse head (forward + backward copies) }. */
void abip_hip_qcp_last_stats(double *out8);
/* the reference's per-phase timers of the last solve (src/abip-qcp/source/abip.c:1084-1093, printed at 1196-1201), seconds, in its order:
   project_lin_sys, solve_barrier_subproblem, calculate res, calculate err_inner, updating work.  The four device phases are sampled
   (hipEvents around one iteration per control read) and scaled to the iteration count; calculate res is the host clock around its calls. */
void abip_hip_qcp_phase_times(double *out5);
/* Column ranges of the sharded conic path (several GPUs): bounds[g] .. bounds[g+1] are rank g's columns (world + 1 entries out), cut behind
 * cones or inside the free / zero / orthant blocks, balanced by non-zeros.  Pure host code.  0 ok; -1 a rotated cone of fewer than 3 entries;
 * -2 fewer blocks than ranks; -3 bad arguments. */
int abip_hip_qcp_dist_partition(const QCPMatrix *A, const QCPCone *K, int world, int *bounds);
/* Pure host code (runs without a GPU; CPU-side parity tests of the host logic): the formulation front end of d->stgs->prob_type and the scaling exactly as
 * abip_qcp applies them, then Ax_out (m) = A x_in and Aty_out (n) = A' y_in with the scaled operator as it is handed to the device (for prob_type 0 / 1 / 3
 * the materialised operator of lasso_config.c:99-128, svm_config.c:177-230, svm_qp_config.c); b_out (m), c_out (n) the scaled right-hand side and cost;
 * scal4 = {sc_b, sc_c, non-zeros of the operator, sparsity flag}; dims2 = {m, n} of the conic problem.  Any output may be NULL.  Returns 0, < 0 on invalid input. */
int abip_hip_qcp_host_probe(const QCPData *d, const QCPCone *K, const double *x_in, const double *y_in, double *Ax_out, double *Aty_out, double *b_out, double *c_out,
                            double *scal4, int *dims2);
/* Unit-level access to the cone kernel: x <- barrier prox of one cone at tmp (soc_barrier_subproblem cones.c:130-161 for kind 0,
 * rsoc_barrier_subproblem cones.c:169-248 for kind 1; the latter reads the incoming x[0], cones.c:183).  0 on success. */
int abip_hip_qcp_cone_prox(int kind, double *x, const double *tmp, double lambda, int len);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
