/*
 * abip_qcp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the generic conic path of leavesgrp/ABIP v2.0.0 (src/abip-qcp, problem type QCP,
 * linsys_solver = 1): scaling, KKT assembly, projection, cone barrier sub-problems, dual update, inner stopping
 * test, residuals, barrier adjustment, solution extraction.  Each routine cites the reference lines it follows.
 *
 * Parity status: PARTIALLY PINNED.  The reference's QCP sources include five MKL headers unconditionally
 * (source/abip.c:13-14, include/cones.h:11-12, include/linsys.h:14-18), so they cannot be compiled here without
 * writing stand-ins for them, which the rules of this build forbid: there is no oracle/_ref for QCP.  What pins
 * this restatement instead (tests/test_qcp_oracle.py):
 *   (1) the reference's own output on the fully literal toy problem of test/test_abip_install.m:32-43, recorded
 *       by the survey from a run of the real code (ipm 10, admm 91, pobj -0.984063813,
 *       dobj -0.984063938, x to 6 digits);
 *   (2) the cross-solver check the reference's test performs (test_abip_install.m:24-27): an LP solved through
 *       this conic path must reach the optimum the pinned LP oracle / the LP reference fixtures reach;
 *   (3) KKT conditions of the returned (x, y, s) on seeded SOCP / QP instances (a property, not a pin);
 *   (4) the cone sub-problems on their own (orc_qcp_cone_prox): on every generic branch the closed forms of cones.c:130-288
 *       satisfy x - t = lambda grad log det(x) to 1e-9, i.e. they are the exact minimisers of the barrier sub-problem the
 *       reference documents -- a check that does not go through any ABIP code.
 * The ordering of the KKT factorisation is our own minimum-degree code instead of AMD (orc_ldl.h).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's shared object.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/abip_qcp.h"
#include "orc_ldl.h"

typedef qcp_int I;
typedef qcp_float F;
typedef long L;

#define MAXF(a, b) (((a) > (b)) ? (a) : (b))
#define MINF(a, b) (((a) < (b)) ? (a) : (b))
#define ABSF(x) (((x) < 0) ? -(x) : (x))
#define MIN_SCALE (1e-3) /* qcp_config.c:2 */
#define MAX_SCALE (1e3)  /* qcp_config.c:3 */
#define EPS_TOL (1E-18)
#define SAFEDIV_POS(X, Y) ((Y) < EPS_TOL ? ((X) / EPS_TOL) : (X) / (Y))

#define ST_SOLVED 1
#define ST_SOLVED_INACC 2
#define ST_UNBOUNDED (-1)
#define ST_INFEASIBLE (-2)
#define ST_FAILED (-4)
#define ST_UNB_INACC (-6)
#define ST_INF_INACC (-7)

/* ---- linalg.c ---------------------------------------------------------------------------------------- */
static F v_dot(const F *x, const F *y, L n) { F s = 0; for (L i = 0; i < n; ++i) s += x[i] * y[i]; return s; }
static F v_nrm2sq(const F *x, L n) { F s = 0; for (L i = 0; i < n; ++i) s += x[i] * x[i]; return s; }
static F v_nrm2(const F *x, L n) { return sqrt(v_nrm2sq(x, n)); }
static F v_nrminf(const F *a, L n) { F mx = 0; for (L i = 0; i < n; ++i) { F t = ABSF(a[i]); if (t >= mx) mx = t; } return mx; }
static F v_mean(const F *x, L n) { F y = 0; for (L i = 0; i < n; ++i) y += x[i]; return y / n; } /* linalg.c:19-31 */
static void sp_accum_A(const QCPMatrix *A, const F *x, F *y) { /* y += A x, linsys.c:242-262 */
  for (I j = 0; j < A->n; ++j) { F xj = x[j]; for (I p = A->p[j]; p < A->p[j + 1]; ++p) y[A->i[p]] += A->x[p] * xj; }
}
static void sp_accum_At(const QCPMatrix *A, const F *x, F *y) { /* y += A' x, linsys.c:191-225 */
  for (I j = 0; j < A->n; ++j) { F yj = y[j]; for (I p = A->p[j]; p < A->p[j + 1]; ++p) yj += A->x[p] * x[A->i[p]]; y[j] = yj; }
}

/* ---- work ---------------------------------------------------------------------------------------------- */
struct OrcLasso; struct OrcSvmqp; struct OrcSvm;
typedef struct {
  int kind;               /* enum problem_type as abip() maps prob_type (abip.c:1341-1348): 0 LASSO (lasso_config.c), 1 SVM-SOCP (svm_config.c), 2 generic QCP (qcp_config.c), 3 SVM-QP (svm_qp_config.c) */
  struct OrcLasso *ls; struct OrcSvmqp *sq; struct OrcSvm *sv;
  I m, n;
  const QCPSettings *stgs;
  QCPMatrix A, Q; int hasQ;
  F *b, *c, *D, *E, *rho_dr;
  F sc_b, sc_c;
  int sparsity;
  /* KKT factor */
  long N; long *P, *Lp, *Li; F *Lx, *Dg, *bp;
  /* iterates (struct ABIP_WORK, abip.h:160-180) */
  F mu, beta, *u, *v, *v_origin, *u_t, *rel_ut, *r, a, nm_inf_b, nm_inf_c;
  /* indirect back-end (linsys_solver = 3) */
  F *Mpre, *Hinv; long tot_cg, cg_solves; F last_Ax_b_norm, last_Qx_norm;
} QW;

typedef struct {
  I last_ipm_iter, last_admm_iter;
  F res_pri, res_dual, rel_gap, res_infeas, res_unbdd, pobj, dobj, tau, kap, res_dif, error_ratio, Ax_b_norm, Qx_ATy_c_s_norm;
} QR;

static void copy_mat(QCPMatrix *dst, const QCPMatrix *src) {
  const I nnz = src->p[src->n];
  dst->m = src->m; dst->n = src->n;
  dst->x = (F *)malloc(sizeof(F) * (nnz > 0 ? nnz : 1)); dst->i = (I *)malloc(sizeof(I) * (nnz > 0 ? nnz : 1)); dst->p = (I *)malloc(sizeof(I) * (src->n + 1));
  memcpy(dst->x, src->x, sizeof(F) * nnz); memcpy(dst->i, src->i, sizeof(I) * nnz); memcpy(dst->p, src->p, sizeof(I) * (src->n + 1));
}

/* cone-wise averaging of the column scale (qcp_config.c:194-212 and its repeats) */
static void cone_average(F *E, const QCPCone *k) {
  I count = 0;
  if (k->q) for (I i = 0; i < k->qsize; ++i) { F me = v_mean(&E[count], k->q[i]); for (I j = 0; j < k->q[i]; ++j) E[j + count] = me; count += k->q[i]; }
  if (k->rq) for (I i = 0; i < k->rqsize; ++i) { F me = v_mean(&E[count], k->rq[i]); for (I j = 0; j < k->rq[i]; ++j) E[j + count] = me; count += k->rq[i]; }
}
/* apply one (D, E) pass to A and Q and fold it into D_hat, E_hat (qcp_config.c:214-262 and its repeats) */
static void apply_pass(QW *w, F *D, F *E, F min_row, F max_row, F min_col, F max_col) {
  const I m = w->m, n = w->n; QCPMatrix *A = &w->A, *Q = &w->Q;
  for (I i = 0; i < m; ++i) { if (D[i] < min_row) D[i] = 1; else if (D[i] > max_row) D[i] = max_row; }
  for (I i = 0; i < n; ++i) {
    if (E[i] < min_col) E[i] = 1; else if (E[i] > max_col) E[i] = max_col;
    for (I j = A->p[i]; j < A->p[i + 1]; ++j) A->x[j] /= E[i];
  }
  if (w->hasQ) {
    for (I i = 0; i < n; ++i) for (I j = Q->p[i]; j < Q->p[i + 1]; ++j) Q->x[j] /= E[i];
    for (I i = 0; i < Q->p[n]; ++i) Q->x[i] /= E[Q->i[i]];
  }
  for (I i = 0; i < A->p[n]; ++i) A->x[i] /= D[A->i[i]];
  for (I i = 0; i < n; ++i) w->E[i] *= E[i];
  for (I i = 0; i < m; ++i) w->D[i] *= D[i];
}

/* the ruiz / origin / pc passes over (A, Q) of a view with w->m rows and w->n columns; D_hat, E_hat accumulate in w->D, w->E
 * (qcp_config.c:130-460; svm_qp_config.c:199-556 runs the same passes over its first n + 1 columns) */
static void scaling_passes(QW *w, const QCPCone *k) {
  const I m = w->m, n = w->n; QCPMatrix *A = &w->A, *Q = &w->Q;
  const F min_row = MIN_SCALE * sqrt((F)n), max_row = MAX_SCALE * sqrt((F)n), min_col = MIN_SCALE * sqrt((F)m), max_col = MAX_SCALE * sqrt((F)m);
  for (I i = 0; i < n; ++i) w->E[i] = 1;
  for (I i = 0; i < m; ++i) w->D[i] = 1;
  F *E = (F *)calloc(n, sizeof(F)), *E1 = (F *)calloc(n, sizeof(F)), *E2 = (F *)calloc(n, sizeof(F)), *D = (F *)calloc(m, sizeof(F));
  if (w->stgs->ruiz_scaling) { /* :158-264 */
    for (int it = 0; it < 10; ++it) {
      memset(E, 0, sizeof(F) * n); memset(E1, 0, sizeof(F) * n); memset(E2, 0, sizeof(F) * n); memset(D, 0, sizeof(F) * m);
      for (I j = 0; j < n; ++j) E1[j] = (A->p[j] == A->p[j + 1]) ? 0 : sqrt(v_nrminf(&A->x[A->p[j]], A->p[j + 1] - A->p[j]));
      if (w->hasQ) for (I j = 0; j < n; ++j) E2[j] = (Q->p[j] == Q->p[j + 1]) ? 0 : sqrt(v_nrminf(&Q->x[Q->p[j]], Q->p[j + 1] - Q->p[j]));
      for (I i = 0; i < n; ++i) E[i] = E1[i] < E2[i] ? E2[i] : E1[i];
      cone_average(E, k);
      for (I i = 0; i < A->p[n]; ++i) if (D[A->i[i]] < ABSF(A->x[i])) D[A->i[i]] = ABSF(A->x[i]);
      for (I i = 0; i < m; ++i) D[i] = sqrt(D[i]);
      apply_pass(w, D, E, min_row, max_row, min_col, max_col);
    }
  }
  if (w->stgs->origin_scaling) { /* :266-363 */
    memset(E, 0, sizeof(F) * n); memset(E1, 0, sizeof(F) * n); memset(E2, 0, sizeof(F) * n); memset(D, 0, sizeof(F) * m);
    for (I i = 0; i < n; ++i) { for (I j = A->p[i]; j < A->p[i + 1]; ++j) E1[i] += A->x[j] * A->x[j]; E1[i] = sqrt(E1[i]); }
    if (w->hasQ) for (I i = 0; i < n; ++i) { for (I j = Q->p[i]; j < Q->p[i + 1]; ++j) E2[i] += Q->x[j] * Q->x[j]; E2[i] = sqrt(E2[i]); }
    for (I i = 0; i < n; ++i) E[i] = sqrt(E1[i] < E2[i] ? E2[i] : E1[i]);
    cone_average(E, k);
    for (I i = 0; i < A->p[n]; ++i) D[A->i[i]] += A->x[i] * A->x[i];
    for (I i = 0; i < m; ++i) D[i] = sqrt(sqrt(D[i]));
    apply_pass(w, D, E, min_row, max_row, min_col, max_col);
  }
  if (w->stgs->pc_scaling) { /* :365-460 with alpha_pc = 1 */
    memset(E, 0, sizeof(F) * n); memset(E1, 0, sizeof(F) * n); memset(E2, 0, sizeof(F) * n); memset(D, 0, sizeof(F) * m);
    for (I i = 0; i < n; ++i) { for (I j = A->p[i]; j < A->p[i + 1]; ++j) E1[i] += pow(ABSF(A->x[j]), 1.0); E1[i] = sqrt(pow(E1[i], 1.0)); }
    if (w->hasQ) for (I i = 0; i < n; ++i) { for (I j = Q->p[i]; j < Q->p[i + 1]; ++j) E2[i] += pow(ABSF(Q->x[j]), 1.0); E2[i] = sqrt(pow(E2[i], 1.0)); }
    for (I i = 0; i < n; ++i) E[i] = E1[i] < E2[i] ? E2[i] : E1[i];
    cone_average(E, k);
    for (I i = 0; i < A->p[n]; ++i) D[A->i[i]] += pow(ABSF(A->x[i]), 1.0);
    for (I i = 0; i < m; ++i) D[i] = sqrt(pow(D[i], 1.0));
    apply_pass(w, D, E, min_row, max_row, min_col, max_col);
  }
  free(E); free(E1); free(E2); free(D);
}
static void scaling_qcp_data(QW *w, const QCPData *d, const QCPCone *k) { /* qcp_config.c:91-491 */
  const I m = w->m, n = w->n;
  memcpy(w->b, d->b, sizeof(F) * m); memcpy(w->c, d->c, sizeof(F) * n);
  scaling_passes(w, k);
  F sc = sqrt(sqrt(v_nrm2sq(w->c, n) + v_nrm2sq(w->b, m))); /* :462-463 (before the division by D_hat / E_hat) */
  for (I i = 0; i < m; ++i) w->b[i] /= w->D[i];
  for (I i = 0; i < n; ++i) w->c[i] /= w->E[i];
  if (sc < MIN_SCALE) sc = 1; else if (sc > MAX_SCALE) sc = MAX_SCALE;
  w->sc_b = 1 / sc; w->sc_c = 1 / sc;
  for (I i = 0; i < m; ++i) w->b[i] *= w->sc_b * w->stgs->scale;
  for (I i = 0; i < n; ++i) w->c[i] *= w->sc_c * w->stgs->scale;
}

/* K = [[-rho_y I, -A], [., Q_upper + rho_x I]] upper triangle (qcp_config.c:699-748), factorised with orc_ldl.h */
static int init_kkt(QW *w) {
  const I m = w->m, n = w->n; const QCPMatrix *A = &w->A, *Q = &w->Q;
  const long N = (long)m + n, cap = (long)m + A->p[n] + (w->hasQ ? Q->p[n] : 0) + n;
  long *Kp = (long *)malloc(sizeof(long) * (N + 1)), *Ki = (long *)malloc(sizeof(long) * cap); F *Kx = (F *)malloc(sizeof(F) * cap);
  long kk = 0;
  for (I i = 0; i < m; ++i) { Kp[i] = kk; Ki[kk] = i; Kx[kk] = -w->rho_dr[i]; ++kk; }
  for (I i = 0; i < n; ++i) {
    Kp[m + i] = kk;
    for (I j = A->p[i]; j < A->p[i + 1]; ++j) { Ki[kk] = A->i[j]; Kx[kk] = -A->x[j]; ++kk; }
    if (!w->hasQ || Q->p[i] == Q->p[i + 1]) { Ki[kk] = m + i; Kx[kk] = w->rho_dr[m + i]; ++kk; }
    else for (I j = Q->p[i]; j < Q->p[i + 1]; ++j) {
      F t;
      if (Q->i[j] > i) continue;                       /* strictly lower entries are zeroed and dropped (:732-745) */
      t = (Q->i[j] == i) ? Q->x[j] + w->rho_dr[m + i] : Q->x[j];
      if (t == 0) continue;                            /* cs_dropzeros */
      Ki[kk] = m + Q->i[j]; Kx[kk] = t; ++kk;
    }
  }
  Kp[N] = kk;
  w->N = N;
  const int rc = orc_ldl_factor(N, Kp, Ki, Kx, &w->P, &w->Lp, &w->Li, &w->Lx, &w->Dg);
  w->bp = (F *)malloc(sizeof(F) * N);
  free(Kp); free(Ki); free(Kx);
  return rc;
}
static void solve_qcp_linsys(QW *w, F *b) { /* qcp_config.c:868-876: negate the first m entries, then K^-1 */
  for (I i = 0; i < w->m; ++i) b[i] *= -1;
  orc_ldl_solve(w->N, w->P, w->Lp, w->Li, w->Lx, w->Dg, b, w->bp);
}

/* ---- indirect back-end (linsys_solver = 3): the DEFINITION abip_amd/csrc/qcp_pcg.h states and justifies -- pcg of linsys.c:629-716 on the
 * y-space Schur complement rho_y I + A H^-1 A', H = rho_x I + Q diagonal.  (Upstream's n-space qcp_pcg, linsys.c:755-851, is unreachable and,
 * restated here first, did not converge within its n-iteration cap: condition number ~1e8 at the default rho_y = 1e-6.) ---- */
static int init_qcp_pcg(QW *w) {
  const I m = w->m, n = w->n;
  w->Hinv = (F *)malloc(sizeof(F) * n); w->Mpre = (F *)malloc(sizeof(F) * m);
  for (I j = 0; j < n; ++j) w->Hinv[j] = w->rho_dr[m + j];
  if (w->hasQ) for (I j = 0; j < n; ++j) for (I q = w->Q.p[j]; q < w->Q.p[j + 1]; ++q) {
    if (w->Q.i[q] == j) w->Hinv[j] += w->Q.x[q]; else if (w->Q.x[q] != 0) return -1; /* a non-diagonal Q is refused */
  }
  for (I j = 0; j < n; ++j) w->Hinv[j] = 1.0 / w->Hinv[j];
  for (I i = 0; i < m; ++i) w->Mpre[i] = w->rho_dr[i];
  for (I j = 0; j < n; ++j) for (I q = w->A.p[j]; q < w->A.p[j + 1]; ++q) w->Mpre[w->A.i[q]] += w->A.x[q] * w->A.x[q] * w->Hinv[j];
  for (I i = 0; i < m; ++i) w->Mpre[i] = 1.0 / w->Mpre[i];
  return 0;
}
static void qcp_G(QW *w, const F *y, F *out, F *tn) { /* out = rho_y y + A H^-1 A' y */
  const I m = w->m, n = w->n;
  memset(tn, 0, sizeof(F) * n);
  sp_accum_At(&w->A, y, tn);
  for (I j = 0; j < n; ++j) tn[j] *= w->Hinv[j];
  for (I i = 0; i < m; ++i) out[i] = y[i] * w->rho_dr[i];
  sp_accum_A(&w->A, tn, out);
}
static I qcp_pcg(QW *w, F *b, const F *y0, I max_iter, F tol) { /* linsys.c:629-716; result overwrites b */
  const I m = w->m, n = w->n;
  F *p = (F *)calloc(m, sizeof(F)), *Gp = (F *)calloc(m, sizeof(F)), *r = (F *)calloc(m, sizeof(F)), *z = (F *)calloc(m, sizeof(F)), *tn = (F *)calloc(n, sizeof(F));
  I it = 0;
  if (!y0) { memcpy(r, b, sizeof(F) * m); memset(b, 0, sizeof(F) * m); }
  else { qcp_G(w, y0, r, tn); for (I i = 0; i < m; ++i) r[i] = b[i] - r[i]; memcpy(b, y0, sizeof(F) * m); }
  if (v_dot(r, r, m) != 0) {
    for (I i = 0; i < m; ++i) z[i] = r[i] * w->Mpre[i];
    F ip = v_dot(z, r, m), ipold;
    memcpy(p, z, sizeof(F) * m);
    while (it < max_iter) {
      qcp_G(w, p, Gp, tn);
      const F alpha = ip / v_dot(p, Gp, m);
      for (I i = 0; i < m; ++i) { b[i] += alpha * p[i]; r[i] -= alpha * Gp[i]; }
      ++it;
      if (sqrt(v_dot(r, r, m)) < tol) break;
      for (I i = 0; i < m; ++i) z[i] = r[i] * w->Mpre[i];
      ipold = ip; ip = v_dot(z, r, m);
      const F beta = ip / ipold;
      for (I i = 0; i < m; ++i) p[i] = z[i] + beta * p[i];
    }
  }
  free(p); free(Gp); free(r); free(z); free(tn);
  return it;
}
/* K z = (-b_y ; b_x) (the system solve_qcp_linsys hands the direct solver, qcp_config.c:868-876) through the Schur complement */
static void solve_qcp_linsys_pcg(QW *w, F *b, const F *warm /* (m+n) or null */, I iter, F tol) {
  const I m = w->m, n = w->n;
  F *t = (F *)malloc(sizeof(F) * n);
  for (I j = 0; j < n; ++j) t[j] = b[m + j] * w->Hinv[j];
  /* g_y = -b_y  =>  rhs = -g_y - A H^-1 g_x = b_y - A H^-1 b_x */
  F *acc = (F *)calloc(m, sizeof(F));
  sp_accum_A(&w->A, t, acc);
  for (I i = 0; i < m; ++i) b[i] = b[i] - acc[i];
  const I its = qcp_pcg(w, b, warm ? warm : 0, m, tol);
  if (iter >= 0) { w->tot_cg += its; w->cg_solves++; }
  memset(t, 0, sizeof(F) * n);
  sp_accum_At(&w->A, b, t);
  for (I j = 0; j < n; ++j) b[m + j] = (b[m + j] + t[j]) * w->Hinv[j];
  free(t); free(acc);
}

/* ==== LASSO reformulation, lasso_config.c ===================================================================================
 * data: X (dm x dn), y (dm), lambda.  Conic form (init_lasso, :8-94): p = dm + 1 rows, q = 2 + dm + 2 dn columns,
 *   x = (x0, x1, z (dm), beta+ (dn), beta- (dn)),  (x0, x1, z) in one rotated cone of dm + 2, beta+- >= 0,
 *   row 0: x0 = 1;  rows 1..dm: z + X beta+ - X beta- = y;  objective 2 x1 + lambda 1'(beta+ + beta-)   [x0 x1 >= |z|^2 / 2].
 * The operator is applied matrix-free (:99-128), the KKT solve goes through the reduced dm x dm / dn x dn system (:506-556, 652-708);
 * the reference factorises that system with QDLDL / CSparse / MKL, this restatement with a dense Cholesky (test sizes only). */
typedef struct OrcLasso {
  I dm, dn; F lambda, sc, sc_b, sc_c, sc_cone1, sc_cone2;
  QCPMatrix X;                 /* scaled copy of the data matrix */
  const QCPMatrix *X0; const F *y0;
  F *D, *E, *D_hat, *data_b;
  I cn; F *chol;               /* lower Cholesky factor (cn x cn, row-major) of the reduced system */
} OrcLasso;

static void lasso_A_times(QW *w, const F *x, F *y) { /* :99-110 */
  OrcLasso *s = w->ls; const I m = s->dm, n = s->dn;
  y[0] += x[0];
  for (I i = 1; i < m + 1; ++i) y[i] += s->D[i - 1] * sqrt(s->sc_cone2) * x[i + 1];
  sp_accum_A(&s->X, &x[m + 2], &y[1]);
  for (I i = 0; i < m; ++i) y[1 + i] *= -1;
  sp_accum_A(&s->X, &x[m + n + 2], &y[1]);
  for (I i = 0; i < m; ++i) y[1 + i] *= -1;
}
static void lasso_AT_times(QW *w, const F *x, F *y) { /* :116-128 */
  OrcLasso *s = w->ls; const I m = s->dm, n = s->dn;
  y[0] += x[0];
  for (I i = 2; i < m + 2; ++i) y[i] += x[i - 1] * s->D[i - 2] * sqrt(s->sc_cone2);
  sp_accum_At(&s->X, &x[1], &y[m + 2]);
  for (I j = 0; j < n; ++j) y[m + 2 + n + j] *= -1;
  sp_accum_At(&s->X, &x[1], &y[m + 2 + n]);
  for (I j = 0; j < n; ++j) y[m + 2 + n + j] *= -1;
}
static void init_lasso(QW *w, const QCPData *d) { /* :8-94 */
  OrcLasso *s = (OrcLasso *)calloc(1, sizeof(OrcLasso));
  w->ls = s; w->kind = 0;
  s->dm = d->m; s->dn = d->n; s->lambda = d->lambda; s->X0 = d->A; s->y0 = d->b;
  w->m = d->m + 1; w->n = 2 + 2 * d->n + d->m;
  w->sparsity = (((F)d->A->p[d->n] / ((F)d->m * d->n)) < 0.1);
  if (w->sparsity) {
    s->sc = 2; s->sc_c = 1 / s->lambda; s->sc_cone2 = s->lambda / d->m * 80;
    s->sc_cone1 = 0.8 / s->sc_c / s->sc_cone2; s->sc_b = s->sc_c * 300 * s->lambda / d->m;
  } else {
    s->sc = d->m < d->n ? 4 : 1;
    s->sc_c = 1 / s->lambda; s->sc_b = s->sc_c; s->sc_cone2 = 0.8; s->sc_cone1 = 1 / s->sc_c;
  }
  s->data_b = (F *)malloc(sizeof(F) * w->m);
  s->data_b[0] = 1; memcpy(&s->data_b[1], d->b, sizeof(F) * d->m);
  s->D = (F *)calloc(d->m, sizeof(F)); s->E = (F *)calloc(d->n, sizeof(F)); s->D_hat = (F *)malloc(sizeof(F) * d->m);
}
static void scaling_lasso_data(QW *w) { /* :133-260 */
  OrcLasso *s = w->ls; const I m = s->dm, n = s->dn;
  copy_mat(&s->X, s->X0);
  QCPMatrix *A = &s->X; F *E = s->E, *D = s->D;
  F avg = 0, avg1 = 0;
  if (w->stgs->scale_E) {
    if (w->sparsity) {
      for (I i = 0; i < n; ++i) { for (I j = A->p[i]; j < A->p[i + 1]; ++j) E[i] += A->x[j] * A->x[j]; avg += sqrt(E[i]); }
      avg /= n;
      for (I i = 0; i < n; ++i) {
        E[i] = avg / sqrt(E[i] + 1e-4) / s->sc;
        if (E[i] > 1000 * sqrt((F)m)) E[i] = 1000 * sqrt((F)m);
        if (E[i] < 0.001 * sqrt((F)m)) E[i] = 1;
        if (E[i] > 50) E[i] = 50;
        avg1 += E[i];
      }
      avg1 /= n;
      for (I i = 0; i < n; ++i) E[i] = avg1 / E[i] / s->sc;
    } else {
      for (I i = 0; i < n; ++i) {
        for (I j = A->p[i]; j < A->p[i + 1]; ++j) E[i] += A->x[j] * A->x[j];
        E[i] = sqrt(E[i]);
        if (E[i] > 1000 * sqrt((F)m)) E[i] = 1000 * sqrt((F)m);
        if (E[i] < 0.001 * sqrt((F)m)) E[i] = 1;
        if (E[i] > 7) E[i] = 7;
        E[i] = 1 / (E[i] * s->sc);
      }
    }
    for (I i = 0; i < n; ++i) for (I j = A->p[i]; j < A->p[i + 1]; ++j) A->x[j] *= E[i];
  }
  for (I q = 0; q < A->p[n]; ++q) D[A->i[q]] += A->x[q] * A->x[q];
  avg = 0;
  for (I i = 0; i < m; ++i) avg += sqrt(2 * D[i] + s->sc_cone2);
  avg /= m;
  for (I i = 0; i < m; ++i) D[i] = avg / sqrt(2 * D[i] + s->sc_cone2);
  for (I q = 0; q < A->p[n]; ++q) A->x[q] *= D[A->i[q]];
  memcpy(w->b, s->data_b, sizeof(F) * w->m);
  w->b[0] = s->sc_cone1;
  for (I i = 1; i < m + 1; ++i) w->b[i] *= D[i - 1];
  for (I i = 0; i < w->m; ++i) w->b[i] *= s->sc_b;
  w->c[0] = 0; w->c[1] = s->sc_cone1 * s->sc_cone2;
  memset(&w->c[2], 0, sizeof(F) * m);
  for (I i = 0; i < n; ++i) { w->c[i + m + 2] = E[i] * s->lambda; w->c[i + m + 2 + n] = E[i] * s->lambda; }
  for (I j = 0; j < w->n; ++j) w->c[j] *= s->sc_c;
  for (I i = 0; i < m; ++i) s->D_hat[i] = D[i] * D[i] * s->sc_cone2 + w->stgs->rho_y;
}
/* form_lasso_kkt (:506-556) as a dense matrix + Cholesky: dm > dn: I/2 + X' D_hat^-1 X (dn x dn), else D_hat + 2 X X' (dm x dm) */
static int init_lasso_linsys(QW *w) {
  OrcLasso *s = w->ls; const I m = s->dm, n = s->dn; const QCPMatrix *A = &s->X;
  const I cn = m > n ? n : m; s->cn = cn;
  F *Xd = (F *)calloc((size_t)m * n, sizeof(F)), *G = (F *)calloc((size_t)cn * cn, sizeof(F));
  for (I j = 0; j < n; ++j) for (I q = A->p[j]; q < A->p[j + 1]; ++q) Xd[(size_t)A->i[q] * n + j] = A->x[q];
  if (m > n) {
    for (I i = 0; i < m; ++i) { const F *row = &Xd[(size_t)i * n]; const F dinv = 1 / s->D_hat[i];
      for (I a = 0; a < n; ++a) { if (row[a] == 0) continue; const F t = row[a] * dinv; for (I bq = 0; bq <= a; ++bq) G[(size_t)a * cn + bq] += t * row[bq]; } }
    for (I a = 0; a < n; ++a) G[(size_t)a * cn + a] += 0.5;
  } else {
    for (I a = 0; a < m; ++a) for (I bq = 0; bq <= a; ++bq) { F t = 0; const F *ra = &Xd[(size_t)a * n], *rb = &Xd[(size_t)bq * n]; for (I j = 0; j < n; ++j) t += ra[j] * rb[j]; G[(size_t)a * cn + bq] = 2 * t; }
    for (I a = 0; a < m; ++a) G[(size_t)a * cn + a] += s->D_hat[a];
  }
  for (I a = 0; a < cn; ++a) { /* in-place lower Cholesky */
    for (I bq = 0; bq <= a; ++bq) {
      F t = G[(size_t)a * cn + bq];
      for (I k = 0; k < bq; ++k) t -= G[(size_t)a * cn + k] * G[(size_t)bq * cn + k];
      if (a == bq) { if (t <= 0) { free(Xd); free(G); return -1; } G[(size_t)a * cn + a] = sqrt(t); }
      else G[(size_t)a * cn + bq] = t / G[(size_t)bq * cn + bq];
    }
  }
  s->chol = G; free(Xd);
  return 0;
}
static void lasso_chol_solve(const OrcLasso *s, F *b) {
  const I cn = s->cn; const F *G = s->chol;
  for (I a = 0; a < cn; ++a) { F t = b[a]; for (I k = 0; k < a; ++k) t -= G[(size_t)a * cn + k] * b[k]; b[a] = t / G[(size_t)a * cn + a]; }
  for (I a = cn - 1; a >= 0; --a) { F t = b[a]; for (I k = a + 1; k < cn; ++k) t -= G[(size_t)k * cn + a] * b[k]; b[a] = t / G[(size_t)a * cn + a]; }
}
static void solve_lasso_linsys(QW *w, F *b) { /* :652-708, the direct branch */
  OrcLasso *s = w->ls; const I m = s->dm, n = s->dn, p = w->m, q = w->n;
  for (I j = 0; j < q; ++j) b[p + j] *= -1;
  F *b2 = (F *)malloc(sizeof(F) * p);
  memcpy(b2, b, sizeof(F) * p);
  lasso_A_times(w, &b[p], b2);
  b[0] = b2[0] / (1 + w->stgs->rho_y);
  if (m > n) {
    for (I i = 0; i < m; ++i) { b2[i + 1] /= s->D_hat[i]; b[i + 1] = b2[i + 1]; }
    F *tmp = (F *)calloc(n, sizeof(F)), *tmp2 = (F *)calloc(m, sizeof(F));
    sp_accum_At(&s->X, &b2[1], tmp);
    lasso_chol_solve(s, tmp);
    sp_accum_A(&s->X, tmp, tmp2);
    for (I i = 0; i < m; ++i) { tmp2[i] /= s->D_hat[i]; b[1 + i] += -tmp2[i]; }
    free(tmp); free(tmp2);
  } else {
    memcpy(&b[1], &b2[1], sizeof(F) * m);
    lasso_chol_solve(s, &b[1]);
  }
  free(b2);
  for (I j = 0; j < q; ++j) b[p + j] *= -1;
  lasso_AT_times(w, b, &b[p]);
}
static void free_lasso(QW *w) {
  OrcLasso *s = w->ls; if (!s) return;
  free(s->X.x); free(s->X.i); free(s->X.p); free(s->D); free(s->E); free(s->D_hat); free(s->data_b); free(s->chol); free(s);
  w->ls = 0;
}

/* ==== SVM as a QP, svm_qp_config.c ==========================================================================================
 * data: X (dm x dn), labels y (+-1), lambda.  QP over x = (w (dn), b, xi (dm), t (dm)), w and b free, xi, t >= 0 (init_svmqp, :8-124):
 *   min 1/2 |w|^2 + 1/(dm lambda) 1'xi   s.t.  diag(y) (X w + b) + xi - t = 1.
 * w->A holds the scaled data block B~ = D^-1 diag(y) [X, 1] E^-1 (dm x (dn + 1)); the +-identity columns are applied on the fly (:129-147).
 * Scaling (:195-590) = the generic passes over the first dn + 1 columns only; residuals and inner test = the generic ones (:150-193, 622-738). */
typedef struct OrcSvmqp {
  I dm, dn; F lambda;
  F *Fd, *H;
  I cn; F *chol;
} OrcSvmqp;
static void svmqp_A_times(QW *w, const F *x, F *y) { /* :129-136 */
  const OrcSvmqp *s = w->sq; const I m = s->dm, n = s->dn;
  sp_accum_A(&w->A, x, y);
  for (I i = 0; i < m; ++i) y[i] += 1 / w->D[i] * (x[n + 1 + i] - x[n + 1 + m + i]);
}
static void svmqp_AT_times(QW *w, const F *x, F *y) { /* :141-147 */
  const OrcSvmqp *s = w->sq; const I m = s->dm, n = s->dn;
  sp_accum_At(&w->A, x, y);
  for (I i = 0; i < m; ++i) { y[n + 1 + i] += 1 / w->D[i] * x[i]; y[n + 1 + m + i] -= 1 / w->D[i] * x[i]; }
}
static void init_svmqp(QW *w, const QCPData *d, F *data_b, F *data_c) { /* :8-124 (the caller's X is not modified here; the reference folds the labels into it in place, :84-86) */
  OrcSvmqp *s = (OrcSvmqp *)calloc(1, sizeof(OrcSvmqp));
  const I m = d->m, n = d->n, q = 1 + n + 2 * m, nnz = d->A->p[n];
  w->sq = s; w->kind = 3; s->dm = m; s->dn = n; s->lambda = d->lambda;
  w->m = m; w->n = q;
  w->sparsity = (((F)nnz / ((F)m * n)) < 0.05);
  w->hasQ = 1;
  w->Q.m = q; w->Q.n = q; w->Q.i = (I *)malloc(sizeof(I) * n); w->Q.x = (F *)malloc(sizeof(F) * n); w->Q.p = (I *)malloc(sizeof(I) * (q + 1));
  for (I i = 0; i < n; ++i) { w->Q.i[i] = i; w->Q.p[i] = i; w->Q.x[i] = 1; }
  for (I i = n; i <= q; ++i) w->Q.p[i] = n;
  for (I i = 0; i < m; ++i) data_b[i] = 1;
  memset(data_c, 0, sizeof(F) * q);
  for (I i = 0; i < m; ++i) data_c[i + n + 1] = 1.0 / (m * s->lambda);
  QCPMatrix *B = &w->A;
  B->m = m; B->n = n + 1; B->p = (I *)malloc(sizeof(I) * (n + 2)); B->i = (I *)malloc(sizeof(I) * (nnz + m)); B->x = (F *)malloc(sizeof(F) * (nnz + m));
  memcpy(B->p, d->A->p, sizeof(I) * (n + 1)); B->p[n + 1] = nnz + m;
  memcpy(B->i, d->A->i, sizeof(I) * nnz);
  for (I k = 0; k < nnz; ++k) B->x[k] = d->A->x[k] * d->b[d->A->i[k]];
  for (I i = 0; i < m; ++i) { B->i[nnz + i] = i; B->x[nnz + i] = d->b[i]; }
  s->Fd = (F *)malloc(sizeof(F) * m); s->H = (F *)malloc(sizeof(F) * q);
}
static void scaling_svmqp_data(QW *w, const F *data_b, const F *data_c, const QCPCone *k) { /* :195-590 */
  OrcSvmqp *s = w->sq; const I m = s->dm, n1 = s->dn + 1, q = w->n;
  memcpy(w->b, data_b, sizeof(F) * m); memcpy(w->c, data_c, sizeof(F) * q);
  for (I i = 0; i < q; ++i) w->E[i] = 1.0; /* :107-110 */
  w->n = n1; scaling_passes(w, k); w->n = q;
  F sc = sqrt(sqrt(v_nrm2sq(w->c, q) + v_nrm2sq(w->b, m)));
  for (I i = 0; i < m; ++i) w->b[i] /= w->D[i];
  for (I i = 0; i < n1; ++i) w->c[i] /= w->E[i];
  if (sc < MIN_SCALE) sc = 1; else if (sc > MAX_SCALE) sc = MAX_SCALE;
  w->sc_b = 1 / sc; w->sc_c = 1 / sc;
  for (I i = 0; i < m; ++i) w->b[i] *= w->sc_b * w->stgs->scale;
  for (I i = 0; i < q; ++i) w->c[i] *= w->sc_c * w->stgs->scale;
  for (I i = 0; i < m; ++i) s->Fd[i] = w->stgs->rho_y + 2 / w->stgs->rho_x / pow(w->D[i], 2);
  for (I i = 0; i < q; ++i) s->H[i] = i < s->dn ? w->stgs->rho_x + w->Q.x[i] : w->stgs->rho_x;
}
/* form_svmqp_kkt (:743-806) dense: dm > dn + 1: diag(H[0:dn+1]) + B~' F^-1 B~, else F + B~ H^-1 B~' */
static int init_svmqp_linsys(QW *w) {
  OrcSvmqp *s = w->sq; const I m = s->dm, n1 = s->dn + 1; const QCPMatrix *B = &w->A;
  const I cn = m > n1 ? n1 : m; s->cn = cn;
  F *Bd = (F *)calloc((size_t)m * n1, sizeof(F)), *G = (F *)calloc((size_t)cn * cn, sizeof(F));
  for (I j = 0; j < n1; ++j) for (I t = B->p[j]; t < B->p[j + 1]; ++t) Bd[(size_t)B->i[t] * n1 + j] = B->x[t];
  if (m > n1) {
    for (I i = 0; i < m; ++i) { const F *row = &Bd[(size_t)i * n1]; const F fi = 1 / s->Fd[i];
      for (I a = 0; a < n1; ++a) { if (row[a] == 0) continue; const F t = row[a] * fi; for (I c2 = 0; c2 <= a; ++c2) G[(size_t)a * cn + c2] += t * row[c2]; } }
    for (I a = 0; a < n1; ++a) G[(size_t)a * cn + a] += s->H[a];
  } else {
    for (I a = 0; a < m; ++a) for (I c2 = 0; c2 <= a; ++c2) { F t = 0; const F *ra = &Bd[(size_t)a * n1], *rb = &Bd[(size_t)c2 * n1]; for (I j = 0; j < n1; ++j) t += ra[j] * rb[j] / s->H[j]; G[(size_t)a * cn + c2] = t; }
    for (I a = 0; a < m; ++a) G[(size_t)a * cn + a] += s->Fd[a];
  }
  for (I a = 0; a < cn; ++a) for (I c2 = 0; c2 <= a; ++c2) {
    F t = G[(size_t)a * cn + c2];
    for (I kk = 0; kk < c2; ++kk) t -= G[(size_t)a * cn + kk] * G[(size_t)c2 * cn + kk];
    if (a == c2) { if (t <= 0) { free(Bd); free(G); return -1; } G[(size_t)a * cn + a] = sqrt(t); } else G[(size_t)a * cn + c2] = t / G[(size_t)c2 * cn + c2];
  }
  s->chol = G; free(Bd);
  return 0;
}
static void dense_chol_solve(I cn, const F *G, F *b) {
  for (I a = 0; a < cn; ++a) { F t = b[a]; for (I k = 0; k < a; ++k) t -= G[(size_t)a * cn + k] * b[k]; b[a] = t / G[(size_t)a * cn + a]; }
  for (I a = cn - 1; a >= 0; --a) { F t = b[a]; for (I k = a + 1; k < cn; ++k) t -= G[(size_t)k * cn + a] * b[k]; b[a] = t / G[(size_t)a * cn + a]; }
}
static void solve_svmqp_linsys(QW *w, F *b) { /* :878-973, the direct branch */
  OrcSvmqp *s = w->sq; const I m = s->dm, n1 = s->dn + 1, p = w->m, q = w->n;
  F *b2 = (F *)malloc(sizeof(F) * p), *tmp = (F *)malloc(sizeof(F) * q);
  memcpy(b2, b, sizeof(F) * p);
  for (I i = 0; i < q; ++i) tmp[i] = b[p + i] / -s->H[i];
  svmqp_A_times(w, tmp, b2);
  if (m > n1) {
    for (I i = 0; i < p; ++i) b2[i] /= s->Fd[i];
    F *t1 = (F *)calloc(n1, sizeof(F)), *t2 = (F *)calloc(m, sizeof(F));
    sp_accum_At(&w->A, b2, t1);
    dense_chol_solve(s->cn, s->chol, t1);
    sp_accum_A(&w->A, t1, t2);
    for (I i = 0; i < m; ++i) { t2[i] /= s->Fd[i]; b2[i] += -t2[i]; }
    free(t1); free(t2);
  } else dense_chol_solve(s->cn, s->chol, b2);
  memcpy(b, b2, sizeof(F) * m);
  free(b2); free(tmp);
  svmqp_AT_times(w, b, &b[p]);
  for (I i = 0; i < q; ++i) b[p + i] /= s->H[i];
}
static void free_svmqp(QW *w) { OrcSvmqp *s = w->sq; if (!s) return; free(s->Fd); free(s->H); free(s->chol); free(s); w->sq = 0; }

/* ==== SVM as an SOCP, svm_config.c ===========================================================================================
 * data: X (dm x dn), labels y, lambda (= C).  x = (x0, x1, r (dn), w+ (dn), b+, w- (dn), b-, xi (dm), t (dm)); (x0, x1, r) in one rotated
 * cone of dn + 2, the rest >= 0; p = dm + dn + 1 rows (init_svm, :8-171):
 *   row 0: x0 = const;  rows 1..dm: diag(y)(X (w+ - w-) + (b+ - b-)) + xi - t = 1;  rows dm+1..: r tied to w+ - w-;  cost x1 + C 1'xi.
 * Operator matrix-free (:177-229); KKT solve by the block elimination of :736-800 with a dense Cholesky for its reduced system.
 * The scale constants of :63-107 are a table of heuristics in (dm, dn, lambda); two of its corners read uninitialised memory in the
 * reference (dm == 10 dn or 10 dm == dn: no branch taken; dm > 10 dn with dn < 10: sc_cone2 never assigned) -- see svm_constants. */
typedef struct OrcSvm {
  I dm, dn; F lambda, sc, sc_b, sc_c, sc_cone1, sc_cone2;
  QCPMatrix A0;          /* data_A = [diag(y) X, y], dm x (dn + 1), un-scaled (:109-133) */
  QCPMatrix A;           /* its scaled copy */
  QCPMatrix wX;          /* first dn columns of A, column i times -2 wE_i (:384-390) */
  F *sc_D, *sc_E, *sc_F, *wy, *wB, *wC, *wD, *wE, *wF, *wG, *wH;
  I cn; F *chol;
} OrcSvm;
static void svm_constants(I m, I n, F lambda, F *sc, F *sc_b, F *sc_c, F *sc_cone1, F *sc_cone2) { /* :63-107 */
  const F l2 = log(2 * lambda) / log(10), l5 = log(5 * lambda) / log(10);
  *sc = 1; *sc_b = 1;
  if ((m < 10 * n) && (10 * m > n)) {
    *sc_c = MAXF(0.45, pow(7.5, -l2) * 2); *sc_cone1 = MAXF(3, l2 * 4 + 4); *sc_cone2 = *sc_cone1;
  } else if (10 * m <= n) { /* (reference: 10 m < n; equality is left undefined there) */
    *sc_cone2 = MAXF(3, l2 * 2 + 2);
    if (lambda >= 1) { *sc_c = MAXF(0.2, pow(0.2, l2) * 7.5); *sc_cone1 = *sc_cone2; }
    else { *sc_c = pow(0.3, l2) * 3; *sc_cone1 = MAXF(0.4, l2 * 0.2 + 0.8); }
  } else { /* m >= 10 n (reference: m > 10 n) */
    if (n < 10) {
      *sc_c = 1 / lambda; *sc_cone1 = 6; *sc_cone2 = 6; /* (sc_cone2 is read before it is ever written in the reference, :85-89: taken as sc_cone1) */
      if (lambda < 0.002) *sc_cone2 = *sc_cone2 - 3 * log(lambda * 500) / log(10);
    } else if (lambda >= 1) { *sc_c = 1 / lambda; *sc_cone1 = 6; *sc_cone2 = lambda; }
    else {
      *sc_c = MINF(pow(5, -l5) * 4, 300); *sc_b = MAXF(0.1, l5 * 0.2 + 0.9); *sc_cone1 = MAXF(0.05, l5 * 0.3 + 0.7); *sc_cone2 = -l5 * 2 + 6;
      if (lambda < 0.002) *sc_cone2 = *sc_cone2 - 3 * log(lambda * 500) / log(10);
    }
  }
}
static void svm_A_times(QW *w, const F *x, F *y) { /* :177-199 */
  const OrcSvm *s = w->sv; const I m = s->dm, n = s->dn;
  y[0] += x[0];
  F *tmp = (F *)malloc(sizeof(F) * (n + 1));
  for (I i = 0; i < n; ++i) tmp[i] = x[n + 2 + i] + (-1) * x[2 * n + 3 + i];
  tmp[n] = 0;
  { QCPMatrix wA = s->A; wA.n = n; sp_accum_A(&wA, tmp, &y[1]); }
  const F db = x[2 * n + 2] - x[3 * n + 3];
  for (I i = 0; i < m; ++i) y[1 + i] += db * s->wy[i];
  for (I i = 0; i < m; ++i) y[i + 1] += (s->wB[i] * x[i + 3 * n + 4] - s->wC[i] * x[i + 3 * n + 4 + m]);
  for (I i = 0; i < n; ++i) y[i + 1 + m] += (s->wD[i] * x[i + 2] - s->wE[i] * tmp[i]);
  free(tmp);
}
static void svm_AT_times(QW *w, const F *x, F *y) { /* :205-229 */
  const OrcSvm *s = w->sv; const I m = s->dm, n = s->dn;
  y[0] += x[0];
  for (I i = 0; i < n; ++i) y[i + 2] += s->wD[i] * x[i + m + 1];
  F *tmp = (F *)malloc(sizeof(F) * n);
  for (I i = 0; i < n; ++i) tmp[i] = -s->wE[i] * x[i + m + 1];
  { QCPMatrix wA = s->A; wA.n = n; sp_accum_At(&wA, &x[1], tmp); }
  for (I i = 0; i < n; ++i) { y[n + 2 + i] += tmp[i]; y[2 * n + 3 + i] += -tmp[i]; }
  const F dt = v_dot(s->wy, &x[1], m);
  y[2 * n + 2] += dt; y[3 * n + 3] -= dt;
  for (I i = 0; i < m; ++i) { y[i + 3 * n + 4] += s->wB[i] * x[i + 1]; y[i + 3 * n + 4 + m] -= s->wC[i] * x[i + 1]; }
  free(tmp);
}
static void init_svm(QW *w, const QCPData *d) { /* :8-171 */
  OrcSvm *s = (OrcSvm *)calloc(1, sizeof(OrcSvm));
  const I m = d->m, n = d->n, nnz = d->A->p[n];
  w->sv = s; w->kind = 1; s->dm = m; s->dn = n; s->lambda = d->lambda;
  w->m = m + n + 1; w->n = 4 + 3 * n + 2 * m;
  w->sparsity = (((F)nnz / ((F)m * n)) < 0.05);
  svm_constants(m, n, s->lambda, &s->sc, &s->sc_b, &s->sc_c, &s->sc_cone1, &s->sc_cone2);
  QCPMatrix *B = &s->A0;
  B->m = m; B->n = n + 1; B->p = (I *)malloc(sizeof(I) * (n + 2)); B->i = (I *)malloc(sizeof(I) * (nnz + m)); B->x = (F *)malloc(sizeof(F) * (nnz + m));
  memcpy(B->p, d->A->p, sizeof(I) * (n + 1)); B->p[n + 1] = nnz + m;
  memcpy(B->i, d->A->i, sizeof(I) * nnz);
  for (I k = 0; k < nnz; ++k) B->x[k] = d->A->x[k] * d->b[d->A->i[k]];
  for (I i = 0; i < m; ++i) { B->i[nnz + i] = i; B->x[nnz + i] = d->b[i]; }
  s->sc_D = (F *)calloc(m, sizeof(F)); s->sc_E = (F *)calloc(n + 1, sizeof(F)); s->sc_F = (F *)calloc(n, sizeof(F));
  s->wy = (F *)malloc(sizeof(F) * m); s->wB = (F *)malloc(sizeof(F) * m); s->wC = (F *)malloc(sizeof(F) * m); s->wF = (F *)malloc(sizeof(F) * m);
  s->wD = (F *)malloc(sizeof(F) * n); s->wE = (F *)malloc(sizeof(F) * n); s->wG = (F *)malloc(sizeof(F) * n); s->wH = (F *)malloc(sizeof(F) * (n + 1));
}
static void scaling_svm_data(QW *w) { /* :281-391 */
  OrcSvm *s = w->sv; const I m = s->dm, n1 = s->dn + 1, n = s->dn;
  copy_mat(&s->A, &s->A0);
  QCPMatrix *A = &s->A; F *E = s->sc_E, *D = s->sc_D;
  F avg = 0;
  if (w->stgs->scale_E) {
    for (I i = 0; i < n1; ++i) { for (I j = A->p[i]; j < A->p[i + 1]; ++j) E[i] += A->x[j] * A->x[j]; E[i] = sqrt(E[i]); avg += E[i]; }
    avg /= n1;
    for (I i = 0; i < n1; ++i) E[i] = avg / E[i];
    for (I i = 0; i < n1; ++i) for (I j = A->p[i]; j < A->p[i + 1]; ++j) A->x[j] *= E[i];
  }
  for (I q = 0; q < A->p[n1]; ++q) D[A->i[q]] += A->x[q] * A->x[q];
  avg = 0;
  for (I i = 0; i < m; ++i) avg += sqrt(D[i]);
  avg /= m;
  for (I i = 0; i < m; ++i) D[i] = avg / sqrt(D[i]);
  for (I q = 0; q < A->p[n1]; ++q) A->x[q] *= D[A->i[q]];
  for (I i = 0; i < n; ++i) s->sc_F[i] = 1 / sqrt(1 + 2 * E[i] * E[i]);
  memset(w->b, 0, sizeof(F) * w->m);
  w->b[0] = s->sc_cone2;
  for (I i = 1; i < m + 1; ++i) w->b[i] = D[i - 1];
  for (I i = 0; i < w->m; ++i) w->b[i] *= s->sc_b;
  memset(w->c, 0, sizeof(F) * w->n);
  w->c[1] = s->sc_c * s->sc_cone1 * s->sc_cone2;
  for (I i = 0; i < m; ++i) w->c[i + 4 + 3 * n] = s->lambda * s->sc_c / s->sc;
  memcpy(s->wy, &A->x[A->p[n]], sizeof(F) * m); /* the label column: dense, rows 0..dm-1 in order */
  for (I i = 0; i < m; ++i) { s->wB[i] = D[i] * (1 / s->sc); s->wC[i] = D[i]; }
  for (I i = 0; i < n; ++i) { s->wD[i] = s->sc_F[i] * -sqrt(s->sc_cone1); s->wE[i] = E[i] * s->sc_F[i]; }
  for (I i = 0; i < m; ++i) s->wF[i] = s->wB[i] * s->wB[i] + s->wC[i] * s->wC[i] + w->stgs->rho_y;
  for (I i = 0; i < n; ++i) s->wG[i] = s->wD[i] * s->wD[i] + 2 * s->wE[i] * s->wE[i] + w->stgs->rho_y;
  for (I i = 0; i < n; ++i) s->wH[i] = 2 - 4 / s->wG[i] * s->wE[i] * s->wE[i];
  s->wH[n] = 2;
  copy_mat(&s->wX, &s->A); s->wX.n = n;
  for (I i = 0; i < n; ++i) for (I j = s->wX.p[i]; j < s->wX.p[i + 1]; ++j) s->wX.x[j] *= (-2 * s->wE[i]);
}
/* form_svm_kkt (:566-634) dense: dm > dn + 1: diag(1 / wH) + A' wF^-1 A ((dn+1)^2), else wF + A diag(wH) A' (dm^2) */
static int init_svm_linsys(QW *w) {
  OrcSvm *s = w->sv; const I m = s->dm, n1 = s->dn + 1; const QCPMatrix *A = &s->A;
  const I cn = m > n1 ? n1 : m; s->cn = cn;
  F *Ad = (F *)calloc((size_t)m * n1, sizeof(F)), *G = (F *)calloc((size_t)cn * cn, sizeof(F));
  for (I j = 0; j < n1; ++j) for (I t = A->p[j]; t < A->p[j + 1]; ++t) Ad[(size_t)A->i[t] * n1 + j] = A->x[t];
  if (m > n1) {
    for (I i = 0; i < m; ++i) { const F *row = &Ad[(size_t)i * n1]; const F fi = 1 / s->wF[i];
      for (I a = 0; a < n1; ++a) { if (row[a] == 0) continue; const F t = row[a] * fi; for (I c2 = 0; c2 <= a; ++c2) G[(size_t)a * cn + c2] += t * row[c2]; } }
    for (I a = 0; a < n1; ++a) G[(size_t)a * cn + a] += 1 / s->wH[a];
  } else {
    for (I a = 0; a < m; ++a) for (I c2 = 0; c2 <= a; ++c2) { F t = 0; const F *ra = &Ad[(size_t)a * n1], *rb = &Ad[(size_t)c2 * n1]; for (I j = 0; j < n1; ++j) t += ra[j] * s->wH[j] * rb[j]; G[(size_t)a * cn + c2] = t; }
    for (I a = 0; a < m; ++a) G[(size_t)a * cn + a] += s->wF[a];
  }
  for (I a = 0; a < cn; ++a) for (I c2 = 0; c2 <= a; ++c2) {
    F t = G[(size_t)a * cn + c2];
    for (I kk = 0; kk < c2; ++kk) t -= G[(size_t)a * cn + kk] * G[(size_t)c2 * cn + kk];
    if (a == c2) { if (t <= 0) { free(Ad); free(G); return -1; } G[(size_t)a * cn + a] = sqrt(t); } else G[(size_t)a * cn + c2] = t / G[(size_t)c2 * cn + c2];
  }
  s->chol = G; free(Ad);
  return 0;
}
static void solve_svm_linsys(QW *w, F *b) { /* :725-806, the direct branch */
  OrcSvm *s = w->sv; const I m = s->dm, n = s->dn, p = w->m, q = w->n;
  for (I j = 0; j < q; ++j) b[p + j] *= -1;
  F *b2 = (F *)malloc(sizeof(F) * p), *b3 = (F *)malloc(sizeof(F) * m), *tmp = (F *)malloc(sizeof(F) * (n + 1));
  memcpy(b2, b, sizeof(F) * p);
  svm_A_times(w, &b[p], b2);
  b[0] = b2[0] / (1 + w->stgs->rho_y);
  for (I i = 0; i < n; ++i) b[i + m + 1] = b2[i + m + 1] / s->wG[i];
  for (I i = 0; i < m; ++i) b3[i] = -b2[1 + i];
  sp_accum_A(&s->wX, &b[m + 1], b3);
  for (I i = 0; i < m; ++i) b3[i] *= -1;
  if (m > n + 1) {
    for (I i = 0; i < m; ++i) b3[i] /= s->wF[i];
    memset(tmp, 0, sizeof(F) * (n + 1));
    sp_accum_At(&s->A, b3, tmp);
    dense_chol_solve(s->cn, s->chol, tmp);
    F *t2 = (F *)calloc(m, sizeof(F));
    sp_accum_A(&s->A, tmp, t2);
    for (I i = 0; i < m; ++i) { t2[i] /= s->wF[i]; b[1 + i] = b3[i] + (-1) * t2[i]; }
    free(t2);
  } else { dense_chol_solve(s->cn, s->chol, b3); memcpy(&b[1], b3, sizeof(F) * m); }
  memset(tmp, 0, sizeof(F) * (n + 1));
  sp_accum_At(&s->wX, &b[1], tmp);
  for (I i = 0; i < n; ++i) { tmp[i] /= s->wG[i]; b[m + 1 + i] += -tmp[i]; }
  free(b2); free(b3); free(tmp);
  for (I j = 0; j < q; ++j) b[p + j] *= -1;
  svm_AT_times(w, b, &b[p]);
}
static void free_mat(QCPMatrix *M) { free(M->x); free(M->i); free(M->p); M->x = 0; M->i = 0; M->p = 0; }
static void free_svm(QW *w) {
  OrcSvm *s = w->sv; if (!s) return;
  free_mat(&s->A0); free_mat(&s->A); free_mat(&s->wX);
  free(s->sc_D); free(s->sc_E); free(s->sc_F); free(s->wy); free(s->wB); free(s->wC); free(s->wD); free(s->wE); free(s->wF); free(s->wG); free(s->wH); free(s->chol); free(s);
  w->sv = 0;
}

/* operator and KKT solve of the formulation in use */
static void op_A(QW *w, const F *x, F *y) { if (w->kind == 0) lasso_A_times(w, x, y); else if (w->kind == 1) svm_A_times(w, x, y); else if (w->kind == 3) svmqp_A_times(w, x, y); else sp_accum_A(&w->A, x, y); }
static void op_At(QW *w, const F *x, F *y) { if (w->kind == 0) lasso_AT_times(w, x, y); else if (w->kind == 1) svm_AT_times(w, x, y); else if (w->kind == 3) svmqp_AT_times(w, x, y); else sp_accum_At(&w->A, x, y); }
static void solve_spe_linsys(QW *w, F *b) { if (w->kind == 0) solve_lasso_linsys(w, b); else if (w->kind == 1) solve_svm_linsys(w, b); else if (w->kind == 3) solve_svmqp_linsys(w, b); else solve_qcp_linsys(w, b); }

/* ---- cones.c:130-288 ------------------------------------------------------------------------------------ */
static void orthant_prox(F *x, const F *t, F lambda, I n) { /* :279-288 */
  for (I i = 0; i < n; ++i) {
    if (t[i] >= 0) x[i] = (t[i] + sqrt(t[i] * t[i] + 4 * lambda)) / 2;
    else x[i] = 2 * lambda / (-t[i] * (1 + sqrt(1 + 4 * lambda / pow(t[i], 2))));
  }
}
static void soc_prox(F *x, const F *tmp, F lambda, I n) { /* :130-161 */
  const F a = tmp[0], tol = 1e-9; const F *b = &tmp[1];
  const F bsq = v_nrm2sq(b, n - 1);
  if (ABSF(a) <= tol) {
    x[0] = sqrt(2 * lambda + bsq / 4);
    for (I i = 0; i < n - 1; ++i) x[1 + i] = b[i] * 0.5;
  } else {
    const F r = 16 * a * a / (8 * lambda - a * a + bsq + sqrt(pow((8 * lambda - a * a + bsq), 2) + 32 * a * a * lambda));
    const F s1 = (r - sqrt(r * (r + 8))) / 2, s2 = (r + sqrt(r * (r + 8))) / 2;
    const F s = a > 0 ? s2 : s1;
    x[0] = (s + 2) * a / s;
    const F sc = (s + 2) / (s + 4);
    for (I i = 0; i < n - 1; ++i) x[1 + i] = b[i] * sc;
  }
}
static void rsoc_prox(F *x, const F *tmp, F lambda, I n) { /* :169-248 */
  const I nx = n - 2; const F ze = tmp[0], zn = tmp[1]; const F *zx = &tmp[2];
  const F xsq = v_nrm2sq(zx, nx);
  if (ze + zn == 0) {
    x[1] = (-ze + sqrt(ze * ze + 4 * lambda + xsq)) / 2;
    x[0] = x[0] + ze; /* sic: reads the previous x[0] (:183) */
    for (I i = 0; i < nx; ++i) x[2 + i] = zx[i] * 0.5;
  } else {
    F w_, s; const F dd = 2 * ze * zn - xsq;
    if (dd < 0) {
      const F q = -dd / (2 * lambda);
      w_ = (2 * pow(ze + zn, 2) / lambda) / q / (1 + 4 / q + sqrt(1 + (4 * (ze * ze + zn * zn + xsq) / lambda + 16) / q / q));
    } else {
      const F q = dd / (2 * lambda);
      w_ = q * (1 - 4 / q + sqrt(1 + (4 * (ze * ze + zn * zn + xsq) / lambda + 16) / q / q)) / 2;
    }
    if (ze + zn > 0) {
      s = (w_ + sqrt(w_ * (w_ + 4))) / 2;
      x[0] = (ze * pow(s + 1, 2) + zn * (s + 1)) / (s * (s + 2));
      x[1] = (zn * pow(s + 1, 2) + ze * (s + 1)) / (s * (s + 2));
      for (I i = 0; i < nx; ++i) x[2 + i] = zx[i] * ((s + 1) / (s + 2));
    } else if (w_ > 10) {
      s = 2 / (w_ + 2 + sqrt(w_ * (w_ + 4)));
      x[0] = (ze * pow(s, 2) + zn * s) / ((s - 1) * (s + 1));
      x[1] = (zn * pow(s, 2) + ze * s) / ((s - 1) * (s + 1));
      for (I i = 0; i < nx; ++i) x[2 + i] = zx[i] * (s / (s + 1));
    } else {
      s = (w_ - sqrt(w_ * (w_ + 4))) / 2;
      x[0] = (ze * pow(s + 1, 2) + zn * (s + 1)) / (s * (s + 2));
      x[1] = (zn * pow(s + 1, 2) + ze * (s + 1)) / (s * (s + 2));
      for (I i = 0; i < nx; ++i) x[2 + i] = zx[i] * ((s + 1) / (s + 2));
    }
  }
}

/* unit-level access for the parity tests: kind 0 SOC, 1 rotated SOC (reads the previous x[0], cones.c:183), 2 orthant */
void orc_qcp_cone_prox(I kind, F *x, const F *tmp, F lambda, I n) {
  if (kind == 0) soc_prox(x, tmp, lambda, n); else if (kind == 1) rsoc_prox(x, tmp, lambda, n); else orthant_prox(x, tmp, lambda, n);
}

/* ---- abip.c ------------------------------------------------------------------------------------------------ */
static void projection(QW *w, I iter) { /* abip.c:186-255 (the direct branch) */
  const I m = w->m, n = w->n; const long mn = (long)m + n;
  F *mu = (F *)malloc(sizeof(F) * mn), *p = (F *)malloc(sizeof(F) * mn), *tem = (F *)malloc(sizeof(F) * mn), *Qp = (F *)calloc(n, sizeof(F));
  for (long i = 0; i < mn; ++i) mu[i] = (w->u[i] + w->v[i]) * w->rho_dr[i];
  const F eta = w->rho_dr[mn] * (w->u[mn] + w->v[mn]);
  memcpy(p, mu, sizeof(F) * mn);
  if (w->stgs->linsys_solver == 3) { /* abip.c:206-224: warm start u + tau r, tolerance from the last residual check (unknown = +inf before the first) */
    F *warm = (F *)malloc(sizeof(F) * mn);
    for (long i = 0; i < mn; ++i) warm[i] = w->u[i] + w->u[mn] * w->r[i];
    F tol = MINF(w->last_Ax_b_norm, w->last_Qx_norm);
    tol = 0.2 * MINF(tol, v_nrminf(warm, n) / pow((F)iter + 1, 1.5));
    tol = MAXF(tol, 1e-12);
    { const char *e = getenv("ORC_QCP_PCG_TOL"); if (e) tol = atof(e); } /* diagnostic: fixed tolerance */
    if (getenv("ORC_QCP_PCG_CHECK") && w->N) { F *p2 = (F *)malloc(sizeof(F) * mn); memcpy(p2, mu, sizeof(F) * mn); solve_qcp_linsys(w, p2);
      F *p3 = (F *)malloc(sizeof(F) * mn); memcpy(p3, mu, sizeof(F) * mn); solve_qcp_linsys_pcg(w, p3, warm, -2, 1e-14);
      F dy = 0, dx = 0; for (long i = 0; i < mn; ++i) { F e = ABSF(p2[i] - p3[i]); if (i < m) dy = MAXF(dy, e); else dx = MAXF(dx, e); }
      fprintf(stderr, "check iter %ld: |dy| %.3e |dx| %.3e (|p| %.3e)\n", (long)iter, dy, dx, v_nrminf(p2, mn)); free(p2); free(p3); }
    solve_qcp_linsys_pcg(w, p, warm, iter, tol);
    free(warm);
  } else solve_spe_linsys(w, p);
  for (long i = 0; i < mn; ++i) tem[i] = p[i] * w->rho_dr[i];
  const F bq = v_dot(w->r, mu, mn) - 2 * v_dot(w->r, tem, mn) - eta;
  if (w->hasQ) sp_accum_A(&w->Q, &p[m], Qp);
  const F cq = -v_dot(&p[m], Qp, n);
  const F a = w->a;
  w->u_t[mn] = (iter > 0) ? (-bq + sqrt(MAXF(0, bq * bq - 4 * a * cq))) / (2 * a) : 1;
  for (long i = 0; i < mn; ++i) w->u_t[i] = p[i] + (-w->u_t[mn]) * w->r[i];
  free(mu); free(p); free(tem); free(Qp);
}
static void solve_barrier_subproblem(QW *w, const QCPCone *c) { /* abip.c:326-413 */
  const I m = w->m, n = w->n; const long l = (long)m + n + 1;
  const F lambda = w->mu / w->beta, al = w->stgs->alpha;
  for (long i = 0; i < l; ++i) w->rel_ut[i] = w->u_t[i] * al + (1 - al) * w->u[i] - w->v[i]; /* :336-342 (scale, axpy, axpy) */
  const F *tmp = w->rel_ut;
  const F tl = tmp[l - 1];
  const F tau = (tl + sqrt(tl * tl + 4 * lambda / w->rho_dr[l - 1])) / 2;
  memcpy(w->u, tmp, sizeof(F) * m);
  w->u[l - 1] = tau;
  I count = 0;
  if (c->qsize && c->q) for (I i = 0; i < c->qsize; ++i) {
    if (c->q[i] == 0) continue;
    if (c->q[i] == 1) orthant_prox(&w->u[m + count], &tmp[m + count], lambda / w->rho_dr[m + count], 1);
    else soc_prox(&w->u[m + count], &tmp[m + count], lambda / w->rho_dr[m + count], c->q[i]);
    count += c->q[i];
  }
  if (c->rqsize && c->rq) for (I i = 0; i < c->rqsize; ++i) {
    if (c->rq[i] < 3) continue; /* sic: count is not advanced (:379-381) */
    rsoc_prox(&w->u[m + count], &tmp[m + count], lambda / w->rho_dr[m + count], c->rq[i]);
    count += c->rq[i];
  }
  if (c->f) { for (I i = 0; i < c->f; ++i) w->u[m + count + i] = tmp[m + count + i]; count += c->f; }
  if (c->z) { for (I i = 0; i < c->z; ++i) w->u[m + count + i] = 0; count += c->z; }
  if (c->l) { orthant_prox(&w->u[m + count], &tmp[m + count], lambda / w->rho_dr[m + count], c->l); count += c->l; }
}
static F svm_inner_conv_check(QW *w) { /* svm_config.c:234-276 */
  const I m = w->m, n = w->n; const long l = (long)m + n + 1;
  const F *y = w->u, *x = &w->u[m], *sv = &w->v_origin[m];
  const F tau = w->u[l - 1], kap = w->v_origin[l - 1];
  F *row1 = (F *)malloc(sizeof(F) * m), *row2 = (F *)malloc(sizeof(F) * n);
  for (I i = 0; i < m; ++i) row1[i] = w->b[i] * -tau;
  svm_A_times(w, x, row1);
  for (I j = 0; j < n; ++j) row2[j] = sv[j] + (-tau) * w->c[j];
  svm_AT_times(w, y, row2);
  for (I j = 0; j < n; ++j) row2[j] *= -1;
  const F last = v_dot(w->b, y, m) - v_dot(w->c, x, n) - kap;
  const F err = sqrt(v_nrm2sq(row1, m) + v_nrm2sq(row2, n) + last * last) / (1 + sqrt(v_nrm2sq(w->u, l) + v_nrm2sq(w->v_origin, l)));
  free(row1); free(row2);
  return err;
}
static F inner_conv_check(QW *w) { /* qcp_config.c:518-557, lasso_config.c:312-353 */
  if (w->kind == 1) return svm_inner_conv_check(w);
  const I m = w->m, n = w->n; const long mn = (long)m + n;
  F *Qu = (F *)malloc(sizeof(F) * (mn + 1)), *Mu = (F *)calloc(mn, sizeof(F));
  op_A(w, &w->u[m], Mu);
  op_At(w, w->u, &Mu[m]);
  for (I j = 0; j < n; ++j) Mu[m + j] *= -1;
  if (w->hasQ) sp_accum_A(&w->Q, &w->u[m], &Mu[m]);
  memcpy(Qu, Mu, sizeof(F) * mn);
  for (I i = 0; i < m; ++i) Qu[i] += -w->u[mn] * w->b[i];
  for (I j = 0; j < n; ++j) Qu[m + j] += w->u[mn] * w->c[j];
  Qu[mn] = -v_dot(w->u, Mu, mn) / w->u[mn] + v_dot(w->u, w->b, m) - v_dot(&w->u[m], w->c, n);
  F num = 0;
  for (long i = 0; i <= mn; ++i) { F t = Qu[i] - w->v_origin[i]; num += t * t; }
  /* qcp_config.c:549-551 normalises by |Qu|, lasso_config.c:343-345 by |u| */
  const F err = sqrt(num) / (1 + v_nrm2(w->kind == 0 ? w->u : Qu, mn + 1) + v_nrm2(w->v_origin, mn + 1));
  free(Qu); free(Mu);
  return err;
}
static void calc_lasso_residuals(QW *w, QR *r, I ipm_iter, I admm_iter) { /* lasso_config.c:358-503 */
  OrcLasso *s = w->ls; const I p = w->m, q = w->n, m = s->dm, n = s->dn;
  r->tau = w->u[p + q];
  const F tau = r->tau;
  F *x = (F *)malloc(sizeof(F) * m), *bp = (F *)malloc(sizeof(F) * n), *bm = (F *)malloc(sizeof(F) * n), *z = (F *)malloc(sizeof(F) * m);
  F *s1 = (F *)malloc(sizeof(F) * n), *s2 = (F *)malloc(sizeof(F) * n), *pr = (F *)malloc(sizeof(F) * m), *dr1 = (F *)malloc(sizeof(F) * n), *dr2 = (F *)malloc(sizeof(F) * n);
  for (I i = 0; i < m; ++i) x[i] = w->u[m + 3 + i] * (sqrt(s->sc_cone2) / (tau * s->sc_b));
  for (I j = 0; j < n; ++j) { bp[j] = w->u[2 * m + 3 + j] * (s->E[j] / (tau * s->sc_b)); bm[j] = w->u[2 * m + 3 + n + j] * (s->E[j] / (tau * s->sc_b)); }
  for (I i = 0; i < m; ++i) z[i] = w->u[1 + i] * (s->D[i] / (tau * s->sc_c));
  for (I j = 0; j < n; ++j) { s1[j] = w->v[2 * m + 3 + j] / (s->E[j] * tau * s->sc_c); s2[j] = w->v[2 * m + n + 3 + j] / (s->E[j] * tau * s->sc_c); }
  memcpy(pr, x, sizeof(F) * m);
  sp_accum_A(s->X0, bp, pr);
  for (I i = 0; i < m; ++i) pr[i] *= -1;
  sp_accum_A(s->X0, bm, pr);
  for (I i = 0; i < m; ++i) pr[i] = -pr[i] - s->y0[i];
  const F this_pr = v_nrm2(pr, m) / MAXF(v_nrm2(s->y0, m), 1);
  memcpy(dr1, s1, sizeof(F) * n);
  sp_accum_At(s->X0, z, dr1);
  for (I j = 0; j < n; ++j) dr1[j] -= s->lambda;
  for (I j = 0; j < n; ++j) dr2[j] = -s2[j];
  sp_accum_At(s->X0, z, dr2);
  for (I j = 0; j < n; ++j) dr2[j] = -dr2[j] - s->lambda;
  const F this_dr = sqrt(v_nrm2sq(dr1, n) + v_nrm2sq(dr2, n)) / (sqrt((F)(2 * n)) * s->lambda);
  F sp = 0, sm = 0;
  for (I j = 0; j < n; ++j) { sp += s->lambda * bp[j]; sm += s->lambda * bm[j]; }
  const F P = 0.5 * v_dot(x, x, m) + sp + sm, zz = v_dot(z, z, m), yz = v_dot(s->y0, z, m);
  const F this_gap = ABSF(P + 0.5 * zz - yz) / (1 + ABSF(P));
  r->last_ipm_iter = ipm_iter; r->last_admm_iter = admm_iter;
  r->dobj = -0.5 * zz + yz; r->pobj = P;
  r->res_dif = MAXF(MAXF(ABSF(this_pr - r->res_pri), ABSF(this_dr - r->res_dual)), ABSF(this_gap - r->rel_gap));
  r->res_pri = this_pr; r->res_dual = this_dr; r->rel_gap = this_gap;
  r->error_ratio = MAXF(r->res_pri / w->stgs->eps_p, MAXF(r->res_dual / w->stgs->eps_d, r->rel_gap / w->stgs->eps_g));
  const F ctu = v_dot(w->c, &w->u[p], q), btu = v_dot(w->b, w->u, p);
  if (ctu < 0) { F *Ax = (F *)calloc(p, sizeof(F)); lasso_A_times(w, &w->u[p], Ax); r->res_unbdd = v_nrm2(Ax, p) / (-ctu); free(Ax); }
  else r->res_unbdd = INFINITY;
  if (btu > 0) { F *t = (F *)calloc(q, sizeof(F)); lasso_AT_times(w, w->u, t); for (I j = 0; j < q; ++j) t[j] += w->v_origin[p + j]; r->res_infeas = v_nrm2(t, q) / btu; free(t); }
  else r->res_infeas = INFINITY;
  free(x); free(bp); free(bm); free(z); free(s1); free(s2); free(pr); free(dr1); free(dr2);
}
static void calc_svm_residuals(QW *w, QR *r, I ipm_iter, I admm_iter) { /* svm_config.c:445-561 */
  OrcSvm *s = w->sv; const I p = w->m, q = w->n, m = s->dm, n = s->dn; const F C = s->lambda;
  r->tau = w->u[p + q];
  const F tau = r->tau;
  F *w1 = (F *)malloc(sizeof(F) * (n + 1)), *xi = (F *)malloc(sizeof(F) * m), *t = (F *)malloc(sizeof(F) * m), *y = (F *)malloc(sizeof(F) * m);
  F *s1 = (F *)malloc(sizeof(F) * m), *s2 = (F *)malloc(sizeof(F) * m), *pr = (F *)malloc(sizeof(F) * m), *BTy = (F *)calloc(n, sizeof(F));
  for (I i = 0; i < n; ++i) w1[i] = (w->u[m + 2 * n + 3 + i] + (-1) * w->u[m + 3 * n + 4 + i]) * (s->sc_E[i] / (tau * s->sc_b));
  w1[n] = (w->u[m + 3 * n + 3] - w->u[m + 4 * n + 4]) / tau * s->sc_E[n] / s->sc_b;
  for (I i = 0; i < m; ++i) {
    xi[i] = w->u[m + 4 * n + 5 + i] * (1 / (tau * s->sc * s->sc_b));
    t[i] = w->u[2 * m + 4 * n + 5 + i] * (1 / (tau * s->sc_b));
    y[i] = w->u[1 + i] / tau * s->sc_D[i] / s->sc_c;
    s2[i] = w->v[2 * m + 4 * n + 5 + i] * (1 / (tau * s->sc_c));
    s1[i] = w->v[m + 4 * n + 5 + i] * (s->sc / (tau * s->sc_c));
  }
  memcpy(pr, xi, sizeof(F) * m);
  sp_accum_A(&s->A0, w1, pr);
  for (I i = 0; i < m; ++i) pr[i] -= (t[i] + 1);
  const F this_pr = v_nrm2(pr, m) / sqrt((F)m);
  F drs = 0;
  for (I i = 0; i < m; ++i) { const F a = y[i] - s2[i], b2 = y[i] + s1[i] - C; drs += a * a; (void)b2; }
  for (I i = 0; i < m; ++i) { const F b2 = y[i] + s1[i] - C; drs += b2 * b2; }
  const F this_dr = sqrt(drs) / (sqrt((F)m) * C);
  { QCPMatrix B = s->A0; B.n = n; sp_accum_At(&B, y, BTy); }
  r->dobj = 0; r->pobj = 0;
  for (I i = 0; i < m; ++i) { r->dobj += y[i]; r->pobj += C * xi[i]; }
  r->pobj += 0.5 * v_dot(w1, w1, n);
  r->dobj -= 0.5 * v_dot(BTy, BTy, n);
  const F this_gap = ABSF(r->dobj - r->pobj) / (1 + ABSF(r->pobj));
  r->last_ipm_iter = ipm_iter; r->last_admm_iter = admm_iter;
  r->res_dif = MAXF(MAXF(ABSF(this_pr - r->res_pri), ABSF(this_dr - r->res_dual)), ABSF(this_gap - r->rel_gap));
  r->res_pri = this_pr; r->res_dual = this_dr; r->rel_gap = this_gap;
  r->error_ratio = MAXF(r->res_pri / w->stgs->eps_p, MAXF(r->res_dual / w->stgs->eps_d, r->rel_gap / w->stgs->eps_g));
  const F ctu = v_dot(w->c, &w->u[p], q), btu = v_dot(w->b, w->u, p);
  if (ctu < 0) { F *Ax = (F *)calloc(p, sizeof(F)); svm_A_times(w, &w->u[p], Ax); r->res_unbdd = v_nrm2(Ax, p) / (-ctu); free(Ax); }
  else r->res_unbdd = INFINITY;
  if (btu > 0) { F *tq = (F *)calloc(q, sizeof(F)); svm_AT_times(w, w->u, tq); for (I j = 0; j < q; ++j) tq[j] += w->v_origin[p + j]; r->res_infeas = v_nrm2(tq, q) / btu; free(tq); }
  else r->res_infeas = INFINITY;
  free(w1); free(xi); free(t); free(y); free(s1); free(s2); free(pr); free(BTy);
}
static void calc_residuals(QW *w, QR *r, I ipm_iter, I admm_iter) { /* qcp_config.c:562-691 */
  if (w->kind == 0) { calc_lasso_residuals(w, r, ipm_iter, admm_iter); return; }
  if (w->kind == 1) { calc_svm_residuals(w, r, ipm_iter, admm_iter); return; }
  const I n = w->n, m = w->m;
  if (admm_iter && r->last_admm_iter == admm_iter) return;
  r->last_ipm_iter = ipm_iter; r->last_admm_iter = admm_iter;
  F *y = (F *)malloc(sizeof(F) * m), *x = (F *)malloc(sizeof(F) * n), *s = (F *)malloc(sizeof(F) * n);
  r->tau = ABSF(w->u[n + m]);
  r->kap = ABSF(w->v_origin[n + m]) / (w->stgs->normalize ? (w->stgs->scale * w->sc_c * w->sc_b) : 1);
  for (I i = 0; i < m; ++i) y[i] = w->u[i] * (1 / r->tau);
  for (I j = 0; j < n; ++j) { x[j] = w->u[m + j] * (1 / r->tau); s[j] = w->v_origin[m + j] * (1 / r->tau); }
  F *Ax = (F *)calloc(m, sizeof(F)), *Ax_b = (F *)malloc(sizeof(F) * m);
  op_A(w, x, Ax);
  for (I i = 0; i < m; ++i) Ax_b[i] = Ax[i] - w->b[i];
  r->Ax_b_norm = v_nrminf(Ax_b, m); w->last_Ax_b_norm = r->Ax_b_norm;
  for (I i = 0; i < m; ++i) { Ax[i] *= w->D[i]; Ax_b[i] *= w->D[i]; }
  const F this_pr = v_nrminf(Ax_b, m) / (w->sc_b + MAXF(v_nrminf(Ax, m), w->sc_b * w->nm_inf_b));
  F *Qx = (F *)calloc(n, sizeof(F)), *ATy = (F *)calloc(n, sizeof(F)), *R = (F *)malloc(sizeof(F) * n);
  F xQx_2 = 0;
  if (w->hasQ) { sp_accum_A(&w->Q, x, Qx); xQx_2 = v_dot(x, Qx, n) / (2 * w->sc_b * w->sc_c); }
  op_At(w, y, ATy);
  for (I j = 0; j < n; ++j) R[j] = Qx[j] - ATy[j] + w->c[j] - s[j];
  r->Qx_ATy_c_s_norm = v_nrminf(R, n); w->last_Qx_norm = r->Qx_ATy_c_s_norm;
  for (I j = 0; j < n; ++j) { Qx[j] *= w->E[j]; ATy[j] *= w->E[j]; R[j] *= w->E[j]; s[j] *= w->E[j]; }
  const F this_dr = v_nrminf(R, n) / (w->sc_c + MAXF(w->sc_c * w->nm_inf_c, v_nrminf(Qx, n)));
  const F cTx = v_dot(w->c, x, n) / (w->sc_b * w->sc_c), bTy = v_dot(w->b, y, m) / (w->sc_b * w->sc_c);
  const F this_gap = ABSF(2 * xQx_2 + cTx - bTy) / (1 + MAXF(2 * xQx_2, MAXF(ABSF(cTx), ABSF(bTy))));
  r->pobj = xQx_2 + cTx; r->dobj = -xQx_2 + bTy;
  r->res_dif = MAXF(MAXF(ABSF(this_pr - r->res_pri), ABSF(this_dr - r->res_dual)), ABSF(this_gap - r->rel_gap));
  r->res_pri = this_pr; r->res_dual = this_dr; r->rel_gap = this_gap;
  r->error_ratio = MAXF(r->res_pri / w->stgs->eps_p, MAXF(r->res_dual / w->stgs->eps_d, r->rel_gap / w->stgs->eps_g));
  const F ctu = v_dot(w->c, &w->u[m], n), btu = v_dot(w->b, w->u, m);
  if (ctu < 0) {
    for (I j = 0; j < n; ++j) Qx[j] *= r->tau;
    for (I i = 0; i < m; ++i) Ax[i] *= r->tau;
    r->res_unbdd = MAXF(v_nrm2(Qx, n), v_nrm2(Ax, m)) / (-ctu);
  } else r->res_unbdd = INFINITY;
  if (btu > 0) {
    for (I j = 0; j < n; ++j) ATy[j] = ATy[j] * r->tau + s[j] * r->tau;
    r->res_infeas = v_nrm2(ATy, n) / btu;
  } else r->res_infeas = INFINITY;
  free(y); free(x); free(s); free(Ax); free(Ax_b); free(Qx); free(ATy); free(R);
}
static I has_converged(const QW *w, const QR *r, I ipm_iter, I admm_iter) { /* abip.c:750-777 */
  const QCPSettings *st = w->stgs;
  if (r->res_pri < st->eps_p && r->res_dual < st->eps_d && r->rel_gap < st->eps_g) return ST_SOLVED;
  if (r->res_dif < st->err_dif * MAXF(MAXF(st->eps_p, st->eps_d), st->eps_g)) return ST_SOLVED_INACC;
  if (r->res_unbdd < st->eps_unb && ipm_iter > 0 && admm_iter > 0) return ST_UNBOUNDED;
  if (r->res_infeas < st->eps_inf && ipm_iter > 0 && admm_iter > 0) return ST_INFEASIBLE;
  return 0;
}
static F adjust_barrier(QW *w, const QR *r) { /* abip.c:994-1071 */
  const QCPSettings *st = w->stgs;
  F sigma = 0.8, gamma;
  const F ratio = w->mu / MINF(MINF(st->eps_p, st->eps_d), st->eps_g);
  if (ratio > 50 && ratio <= 100) gamma = 1.5;
  else if (ratio > 10 && ratio <= 50) gamma = 1.3;
  else if (ratio > 5 && ratio <= 10) gamma = 1.2;
  else if (ratio > 1 && ratio <= 5) gamma = 1.1;
  else if (ratio > 0.5 && ratio <= 1) gamma = 1;
  else if (ratio > 0.05 && ratio <= 0.5) gamma = 0.9;
  else if (ratio > 0.005 && ratio <= 0.05) gamma = 0.8;
  else if (ratio > 0.0005 && ratio <= 0.005) gamma = 0.7;
  else if (ratio > 0.00005 && ratio <= 0.0005) gamma = 0.6;
  else gamma = 0.5;
  const F mr = r->error_ratio;
  if (mr > 22) gamma *= 4.4;
  else if (mr > 18 && mr <= 22) gamma *= 4.2;
  else if (mr > 15 && mr <= 18) gamma *= 4;
  else if (mr > 12 && mr <= 15) gamma *= 3.8;
  else if (mr > 8 && mr <= 12) gamma *= 3.6;
  else if (mr > 6 && mr <= 8) { sigma = 0.81; gamma *= 3.4; }
  else if (mr > 4 && mr <= 6) { sigma = 0.82; gamma *= 3.4; }
  else if (mr > 3 && mr <= 4) { sigma = 0.83; gamma *= 3.2; }
  else if (mr > 2 && mr <= 3) { sigma = 0.85; gamma *= 2.8; }
  else if (mr > 1.5 && mr <= 2) { sigma = 0.85; gamma *= 2.6; }
  else if (mr < 1.5) { sigma = 0.85; gamma *= 2.4; }
  sigma *= 0.2;
  w->mu = sigma * w->mu;
  return gamma * pow(w->mu, st->psi);
}
static void get_solution(QW *w, QCPSolution *sol, QCPInfo *info, const QR *r, I ipm_iter, I admm_iter) { /* abip.c:559-587 */
  const I m = w->m, n = w->n;
  if (!sol->x) sol->x = (F *)malloc(sizeof(F) * n);
  if (!sol->y) sol->y = (F *)malloc(sizeof(F) * m);
  if (!sol->s) sol->s = (F *)malloc(sizeof(F) * n);
  memcpy(sol->x, &w->u[m], sizeof(F) * n); memcpy(sol->y, w->u, sizeof(F) * m); memcpy(sol->s, &w->v[m], sizeof(F) * n);
  const I sv = info->status_val;
  if (sv == 0 || sv == ST_SOLVED || sv == ST_SOLVED_INACC) { /* solved(), abip.c:427-443 */
    const F sc = SAFEDIV_POS(1.0, r->tau);
    for (I j = 0; j < n; ++j) { sol->x[j] *= sc; sol->s[j] *= sc; }
    for (I i = 0; i < m; ++i) sol->y[i] *= sc;
    if (sv == 0 || sv == 2) { strcpy(info->status, "Solved/Inaccurate"); info->status_val = ST_SOLVED_INACC; }
    else { strcpy(info->status, "Solved"); info->status_val = ST_SOLVED; }
  } else if (sv == ST_INFEASIBLE || sv == ST_INF_INACC) { /* abip.c:480-495 */
    const F bty = r->dobj * r->tau;
    for (I i = 0; i < m; ++i) sol->y[i] *= 1 / bty;
    for (I j = 0; j < n; ++j) { sol->s[j] *= 1 / bty; sol->x[j] = NAN; }
    strcpy(info->status, "Infeasible"); info->status_val = ST_INFEASIBLE;
  } else { /* abip.c:497-512 */
    const F ctx = r->pobj * r->tau;
    for (I j = 0; j < n; ++j) { sol->x[j] *= -1 / ctx; sol->s[j] = NAN; }
    for (I i = 0; i < m; ++i) sol->y[i] = NAN;
    strcpy(info->status, "Unbounded"); info->status_val = ST_UNBOUNDED;
  }
  if (w->stgs->normalize && w->kind == 0) { /* un_scaling_lasso_sol, lasso_config.c:296-311: beta = E o (beta+ - beta-) / sc_b replaces x (y and s are dropped) */
    const OrcLasso *ls = w->ls;
    F *beta = (F *)malloc(sizeof(F) * ls->dn);
    for (I j = 0; j < ls->dn; ++j) beta[j] = (sol->x[ls->dm + 2 + j] + (-1) * sol->x[ls->dm + ls->dn + 2 + j]) * ls->E[j] * (1 / ls->sc_b);
    memcpy(sol->x, beta, sizeof(F) * ls->dn);
    free(beta);
  } else if (w->stgs->normalize && w->kind == 1) { /* un_scaling_svm_sol, svm_config.c:410-440 */
    const OrcSvm *sv = w->sv; const I dn = sv->dn, dm = sv->dm;
    F *wv = (F *)malloc(sizeof(F) * dn), *xi = (F *)malloc(sizeof(F) * dm);
    for (I j = 0; j < dn; ++j) wv[j] = (sol->x[dn + 2 + j] + (-1) * sol->x[2 * dn + 3 + j]) * sv->sc_E[j] * (1 / sv->sc_b);
    const F bb = (sol->x[2 * dn + 2] - sol->x[3 * dn + 3]) * sv->sc_E[dn] / sv->sc_b;
    for (I i = 0; i < dm; ++i) xi[i] = sol->x[3 * dn + 4 + i] * (1 / (sv->sc_b * sv->sc_c));
    memcpy(sol->x, wv, sizeof(F) * dn); sol->y[0] = bb; memcpy(sol->s, xi, sizeof(F) * dm);
    free(wv); free(xi);
  } else if (w->stgs->normalize && w->kind == 3) { /* un_scaling_svmqp_sol, svm_qp_config.c:595-619: x / (E sc_b) -> w = x[0:dn], b = x[dn], xi = x[dn+1 : dn+1+dm] */
    const OrcSvmqp *sq = w->sq;
    for (I j = 0; j < n; ++j) sol->x[j] /= (w->E[j] * w->sc_b);
    sol->y[0] = sol->x[sq->dn];
    memcpy(sol->s, &sol->x[sq->dn + 1], sizeof(F) * sq->dm);
  } else if (w->stgs->normalize) { /* un_scaling_qcp_sol, qcp_config.c:496-513 */
    for (I j = 0; j < n; ++j) sol->x[j] /= (w->E[j] * w->sc_b);
    for (I i = 0; i < m; ++i) sol->y[i] /= (w->D[i] * w->sc_c);
    for (I j = 0; j < n; ++j) sol->s[j] *= w->E[j] / (w->sc_c * w->stgs->scale);
  }
  info->ipm_iter = ipm_iter + 1; info->admm_iter = admm_iter; /* get_info, abip.c:526-557 */
  info->res_infeas = r->res_infeas; info->res_unbdd = r->res_unbdd;
  if (info->status_val == ST_SOLVED || info->status_val == ST_SOLVED_INACC) {
    info->rel_gap = r->rel_gap; info->res_pri = r->res_pri; info->res_dual = r->res_dual; info->pobj = r->pobj; info->dobj = r->dobj;
  } else if (info->status_val == ST_UNBOUNDED) { info->rel_gap = info->res_pri = info->res_dual = NAN; info->pobj = info->dobj = -INFINITY; }
  else { info->rel_gap = info->res_pri = info->res_dual = NAN; info->pobj = info->dobj = INFINITY; }
}

void orc_qcp_set_default_settings(QCPData *d) { /* util.c:203-255 */
  QCPSettings *s = d->stgs;
  const double nz = d->A ? d->A->p[d->n] : 0, sparsity = nz / ((double)d->m * d->n);
  s->normalize = 1; s->scale_E = 1; s->scale_bc = 1; s->max_ipm_iters = 500; s->max_admm_iters = 10000000;
  s->eps = s->eps_p = s->eps_d = s->eps_g = s->eps_inf = s->eps_unb = 1e-3; s->alpha = 1.8; s->cg_rate = 2.0;
  s->use_indirect = 0; s->scale = 1.0; s->rho_y = 1e-6; s->rho_x = 1; s->rho_tau = 1; s->verbose = 1; s->err_dif = 0;
  s->inner_check_period = 500; s->outer_check_period = 1;
  s->linsys_solver = ((double)d->m * d->n > 1e12) ? 3 : (sparsity > 0.4 ? 5 : 1);
  s->prob_type = 2; /* the mex overrides the default 3 with the enum value QCP = 2 (abip_qcp_mex.c:436) */
  s->time_limit = INFINITY; s->psi = 1; s->origin_scaling = 1; s->ruiz_scaling = 1; s->pc_scaling = 0;
}

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; }

/* optional trace of (u, v, u_t) after each of the first T inner iterations */
static F *g_trace = 0; static I g_trace_T = 0, g_trace_n = 0;
/* test hook: the oracle's own LDL' (orc_ldl.h) on any symmetric quasi-definite K (upper triangle, CSC): b <- K^-1 b.  Held against the reference's
 * QDLDL by tests/test_qdldl_pin_cpu.py -- the one piece of the conic path that can be pinned against reference code in this image. */
int orc_ldl_solve_upper(int N, const int *Kp, const int *Ki, const double *Kx, double *b) {
  orc_I *P = NULL, *Lp = NULL, *Li = NULL; orc_F *Lx = NULL, *Dg = NULL;
  orc_I *kp = (orc_I *)malloc(sizeof(orc_I) * ((size_t)N + 1)), *ki = (orc_I *)malloc(sizeof(orc_I) * (size_t)(Kp[N] > 0 ? Kp[N] : 1));
  for (int j = 0; j <= N; ++j) kp[j] = Kp[j];
  for (int q = 0; q < Kp[N]; ++q) ki[q] = Ki[q];
  const int rc = orc_ldl_factor(N, kp, ki, Kx, &P, &Lp, &Li, &Lx, &Dg);
  if (rc == 0) {
    orc_F *bp = (orc_F *)malloc(sizeof(orc_F) * N);
    orc_ldl_solve(N, P, Lp, Li, Lx, Dg, b, bp);
    free(bp);
  }
  free(kp); free(ki); free(P); free(Lp); free(Li); free(Lx); free(Dg);
  return rc;
}
void orc_qcp_set_trace(I T, F *buf) { g_trace = buf; g_trace_T = T; g_trace_n = 0; }
I orc_qcp_trace_count(void) { return g_trace_n; }

/* test hook: stop after the formulation + scaling and hand out two products with the scaled operator (matrix-free for prob_type 0 / 1 / 3, as the reference
   applies it), the scaled b and c and {sc_b, sc_c}: the CPU-side check of the product's host front ends (abip_hip_qcp_host_probe) */
static const F *g_px = 0, *g_py = 0; static F *g_pAx = 0, *g_pAty = 0, *g_pb = 0, *g_pc = 0, *g_ps = 0; static int g_probe = 0;
void orc_qcp_set_probe(int on, const F *x, const F *y, F *Ax, F *Aty, F *b, F *c, F *scal2) { g_probe = on; g_px = x; g_py = y; g_pAx = Ax; g_pAty = Aty; g_pb = b; g_pc = c; g_ps = scal2; }

qcp_int orc_qcp_solve(const QCPData *d, QCPSolution *sol, QCPInfo *info, QCPCone *K) { /* abip(), abip.c:1335-1371 */
  const int lasso = d && d->stgs && d->stgs->prob_type == 0; /* abip.c:1341-1348: 0 LASSO, 1 SVM, 2 QCP, 3 SVMQP */
  const int svmqp = d && d->stgs && d->stgs->prob_type == 3, svm = d && d->stgs && d->stgs->prob_type == 1;
  const int ml = lasso || svmqp || svm;
  if (!d || !sol || !info || !K || !d->A || !d->b || (!ml && !d->c) || (d->stgs->linsys_solver != 1 && d->stgs->linsys_solver != 3) ||
      (d->stgs->prob_type != 2 && !ml) || (ml && (d->stgs->linsys_solver != 1 || !(d->lambda > 0)))) {
    if (info) { info->status_val = ST_FAILED; strcpy(info->status, "Failure"); }
    return ST_FAILED;
  }
  const double t_init = now_ms();
  QW W; memset(&W, 0, sizeof(W)); QW *w = &W;
  w->kind = 2; w->stgs = d->stgs;
  F *sq_b = 0, *sq_c = 0;
  if (lasso) init_lasso(w, d);
  else if (svm) init_svm(w, d);
  else if (svmqp) { sq_b = (F *)malloc(sizeof(F) * d->m); sq_c = (F *)malloc(sizeof(F) * (1 + d->n + 2 * d->m)); init_svmqp(w, d, sq_b, sq_c); }
  else { w->m = d->m; w->n = d->n; }
  const I m = w->m, n = w->n; const long l = (long)m + n + 1;
  if (!ml) w->hasQ = d->Q != 0;
  { /* validate, abip.c:779-832 + cones.c:37-81 */
    long dims = K->l + K->z + K->f;
    for (I i = 0; K->q && i < K->qsize; ++i) dims += K->q[i];
    for (I i = 0; K->rq && i < K->rqsize; ++i) dims += K->rq[i];
    const QCPSettings *s = d->stgs;
    if (n <= 0 || m > n || dims != n || s->max_ipm_iters <= 0 || s->max_admm_iters <= 0 || s->eps_p <= 0 || s->eps_d <= 0 || s->eps_g <= 0 ||
        s->eps_inf <= 0 || s->eps_unb <= 0 || s->alpha <= 0 || s->alpha >= 2 || s->rho_y <= 0) {
      info->status_val = ST_FAILED; strcpy(info->status, "Failure"); return ST_FAILED;
    }
  }
  if (!ml) w->sparsity = ((d->A->p[n] / (m * n)) < 0.05); /* integer division, qcp_config.c:22 */
  w->rho_dr = (F *)malloc(sizeof(F) * l);
  for (long i = 0; i < l; ++i) w->rho_dr[i] = i < m ? d->stgs->rho_y : (i < m + n ? d->stgs->rho_x : d->stgs->rho_tau);
  if (!ml) copy_mat(&w->A, d->A);
  if (w->hasQ && !ml) copy_mat(&w->Q, d->Q);
  w->b = (F *)malloc(sizeof(F) * m); w->c = (F *)malloc(sizeof(F) * n); w->D = (F *)malloc(sizeof(F) * m); w->E = (F *)malloc(sizeof(F) * n);
  w->mu = 1.0; w->beta = 1.0;
  w->u = (F *)calloc(l, sizeof(F)); w->v = (F *)calloc(l, sizeof(F)); w->v_origin = (F *)calloc(l, sizeof(F)); w->u_t = (F *)calloc(l, sizeof(F));
  w->rel_ut = (F *)calloc(l, sizeof(F)); w->r = (F *)calloc(l, sizeof(F));
  if (lasso) scaling_lasso_data(w); /* (nm_inf_b / nm_inf_c, abip.c:875-876, are not used by the LASSO residuals) */
  else if (svm) scaling_svm_data(w);
  else if (svmqp) { w->nm_inf_b = v_nrminf(sq_b, m); w->nm_inf_c = v_nrminf(sq_c, n); scaling_svmqp_data(w, sq_b, sq_c, K); free(sq_b); free(sq_c); }
  else { w->nm_inf_b = v_nrminf(d->b, m); w->nm_inf_c = v_nrminf(d->c, n); scaling_qcp_data(w, d, K); }
  w->last_Ax_b_norm = INFINITY; w->last_Qx_norm = INFINITY;
  if (g_probe) {
    if (g_pAx && g_px) { memset(g_pAx, 0, sizeof(F) * m); op_A(w, g_px, g_pAx); }
    if (g_pAty && g_py) { memset(g_pAty, 0, sizeof(F) * n); op_At(w, g_py, g_pAty); }
    if (g_pb) memcpy(g_pb, w->b, sizeof(F) * m);
    if (g_pc) memcpy(g_pc, w->c, sizeof(F) * n);
    if (g_ps) { g_ps[0] = w->sc_b; g_ps[1] = w->sc_c; }
    info->status_val = 0; strcpy(info->status, "Probe");
    return 0;
  }
  if (d->stgs->linsys_solver == 3) { if (init_qcp_pcg(w) < 0) { info->status_val = ST_FAILED; strcpy(info->status, "Failure"); return ST_FAILED; } if (getenv("ORC_QCP_PCG_CHECK")) init_kkt(w); }
  else if ((lasso ? init_lasso_linsys(w) : svm ? init_svm_linsys(w) : svmqp ? init_svmqp_linsys(w) : init_kkt(w)) < 0) { info->status_val = ST_FAILED; strcpy(info->status, "Failure"); return ST_FAILED; }
  info->setup_time = now_ms() - t_init;
  const double t0 = now_ms();
  QR R; memset(&R, 0, sizeof(R)); QR *r = &R;
  info->status_val = 0;
  r->last_ipm_iter = -1; r->last_admm_iter = -1; r->res_pri = r->res_dual = r->rel_gap = r->error_ratio = 1e8;
  F tol_inner = 4 * pow(w->mu, d->stgs->psi);
  { /* update_work, abip.c:912-992 */
    F *x = &w->u[m]; I count = 0;
    for (I i = 0; K->q && i < K->qsize; ++i) { if (K->q[i] == 0) continue; memset(&x[count], 0, sizeof(F) * K->q[i]); x[count] = 1; count += K->q[i]; }
    for (I i = 0; K->rq && i < K->rqsize; ++i) { if (K->rq[i] < 3) continue; memset(&x[count], 0, sizeof(F) * K->rq[i]); x[count] = 1; x[count + 1] = 1; count += K->rq[i]; }
    for (I i = 0; i < K->f + K->z; ++i) x[count + i] = 0;
    count += K->f + K->z;
    for (I i = 0; i < K->l; ++i) x[count + i] = 1;
    w->u[m + n] = 1.0;
    memcpy(w->v, w->u, sizeof(F) * l);
    /* pre_calculate, abip.c:886-910 */
    for (I i = 0; i < m; ++i) w->r[i] = -w->b[i];
    memcpy(&w->r[m], w->c, sizeof(F) * n);
    if (d->stgs->linsys_solver == 3) solve_qcp_linsys_pcg(w, w->r, 0, -1, 1e-12); /* abip.c:899 */
    else solve_spe_linsys(w, w->r);
    F acc = 0; for (long i = 0; i < (long)m + n; ++i) acc += (w->r[i] * w->rho_dr[i]) * w->r[i];
    w->a = w->rho_dr[m + n] + acc;
  }
  I i, j = 0, k = 0;
  const QCPSettings *st = d->stgs;
  for (i = 0; i < st->max_ipm_iters; ++i) {
    for (j = 0; j < st->max_admm_iters; ++j) {
      projection(w, k);
      solve_barrier_subproblem(w, K);
      for (long q = 0; q < l; ++q) w->v[q] = w->u[q] - w->rel_ut[q]; /* update_dual_vars, abip.c:314-324 */
      for (long q = 0; q < l; ++q) w->v_origin[q] = w->v[q] * w->rho_dr[q];
      if (g_trace && g_trace_n < g_trace_T) { F *dst = g_trace + 3 * l * g_trace_n; memcpy(dst, w->u, sizeof(F) * l); memcpy(dst + l, w->v, sizeof(F) * l); memcpy(dst + 2 * l, w->u_t, sizeof(F) * l); g_trace_n++; }
      k += 1;
      const F err_inner = inner_conv_check(w);
      if (err_inner < tol_inner) break;
      if ((j + 1) % st->inner_check_period == 0 || r->error_ratio <= 8) {
        calc_residuals(w, r, i, k);
        if ((info->status_val = has_converged(w, r, i, k)) != 0 || (double)k + 1 >= (double)st->max_admm_iters * st->max_ipm_iters || i + 1 >= st->max_ipm_iters) {
          get_solution(w, sol, info, r, i, k); info->solve_time = now_ms() - t0; goto done;
        }
      }
    }
    if (w->sparsity || (i + 1) % st->outer_check_period == 0) {
      calc_residuals(w, r, i, k);
      if ((info->status_val = has_converged(w, r, i, k)) != 0 || (double)k + 1 >= (double)st->max_admm_iters * st->max_ipm_iters || i + 1 >= st->max_ipm_iters) {
        get_solution(w, sol, info, r, i, k); info->solve_time = now_ms() - t0; goto done;
      }
    }
    tol_inner = adjust_barrier(w, r);
  }
done:
  info->avg_linsys_time = 0; info->avg_cg_iters = w->cg_solves ? (F)w->tot_cg / (F)w->cg_solves : 0;
  free(w->Mpre); free(w->Hinv);
  free_lasso(w); free_svmqp(w); free_svm(w);
  free(w->rho_dr); free(w->A.x); free(w->A.i); free(w->A.p);
  if (w->hasQ) { free(w->Q.x); free(w->Q.i); free(w->Q.p); }
  free(w->b); free(w->c); free(w->D); free(w->E); free(w->u); free(w->v); free(w->v_origin); free(w->u_t); free(w->rel_ut); free(w->r);
  free(w->P); free(w->Lp); free(w->Li); free(w->Lx); free(w->Dg); free(w->bp);
  return info->status_val;
}
