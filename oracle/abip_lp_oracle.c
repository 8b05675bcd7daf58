/*
 * abip_lp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the ABIP-LP solver path of
 * leavesgrp/ABIP v2.0.0 (src/abip-lp).  It exists so that the HIP path can be
 * compared against the reference's algorithm on a machine that has no copy of
 * the reference (the GPU box).  Every routine cites the reference file:line it
 * follows; the arithmetic order of every floating-point sum follows the
 * reference so that, on the same input, this oracle reproduces the reference's
 * iterates to rounding (indirect back-end: bit-for-bit in practice; direct
 * back-end: the fill-reducing ordering is our own minimum-degree code instead of
 * SuiteSparse AMD, so the LDL' factors differ in elimination order and the
 * solves agree to ~1e-12 relative, not bitwise).
 *
 * Parity status: PINNED against the real reference (oracle/_ref, built by
 * oracle/Makefile from /root/reference) by tests/test_oracle_vs_ref.py and by
 * the committed golden fixtures in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object.
 */
#define _POSIX_C_SOURCE 200809L
#include "abip_lp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "orc_ldl.h"

typedef abip_int I;
typedef abip_float F;

#define MAXF(a, b) (((a) > (b)) ? (a) : (b))
#define MINF(a, b) (((a) < (b)) ? (a) : (b))
#define ABSF(x) (((x) < 0) ? -(x) : (x))
#define EPS_TOL (1E-18)                                                /* glbopts.h:157 */
#define SAFEDIV_POS(X, Y) ((Y) < EPS_TOL ? ((X) / EPS_TOL) : (X) / (Y)) /* glbopts.h:158 */
#define INDETERMINATE_TOL (1e-9)                                       /* glbopts.h:161 */
#define MIN_SCALE (1e-3)                                               /* common.c:4    */
#define MAX_SCALE (1e3)                                                /* common.c:5    */

/* ------------------------------------------------------------------------- */
/* dense vector helpers -- src/abip-lp/src/linalg.c (sequential sums)         */
/* ------------------------------------------------------------------------- */
static void v_scale(F *a, F b, I len) { for (I i = 0; i < len; ++i) a[i] *= b; }            /* linalg.c:61-73  */
static void v_axpy(F *a, const F *b, I len, F sc) { for (I i = 0; i < len; ++i) a[i] += sc * b[i]; } /* :236-249 */
static F v_dot(const F *x, const F *y, I len) { F s = 0.0; for (I i = 0; i < len; ++i) s += x[i] * y[i]; return s; } /* :78-92 */
static F v_nrm2sq(const F *v, I len) { F s = 0.0; for (I i = 0; i < len; ++i) s += v[i] * v[i]; return s; }          /* :97-110 */
static F v_nrm2(const F *v, I len) { return sqrt(v_nrm2sq(v, len)); }                                               /* :115-122 */
static F v_nrm1(const F *v, I len) { F s = 0.0; for (I i = 0; i < len; ++i) s += ABSF(v[i]); return s; }             /* :149-162 */
static F v_nrminf(const F *a, I len) { F mx = 0.0; for (I i = 0; i < len; ++i) { F t = ABSF(a[i]); if (t >= mx) mx = t; } return mx; } /* :183-201 */
static F v_minabs_sqrt(const F *a, I len, F ref) { /* linalg.c:126-144 */
  for (I i = 0; i < len; ++i) { F t = ABSF(a[i]); if (t <= ref && t > 0) ref = t; }
  return sqrt(ref);
}

/* ------------------------------------------------------------------------- */
/* sparse matrix-vector products -- linsys/common.c                           */
/* ------------------------------------------------------------------------- */
/* The reference carries one OpenMP pragma on the path, on this loop (common.c:620-622).  Its own build must not define _OPENMP
 * (the pragma's private list is broken), so the multi-core CPU baseline is this restatement built with
 * -fopenmp (liboracle_lp_omp.so): every y[j] is still one sequential sum, so the result is bit-identical to the serial build. */
#ifdef _OPENMP
#include <omp.h>
void orc_set_threads(int t) { omp_set_num_threads(t > 0 ? t : 1); }
int orc_get_threads(void) { return omp_get_max_threads(); }
#else
void orc_set_threads(int t) { (void)t; }
int orc_get_threads(void) { return 1; }
#endif
void orc_accum_by_Atrans(I n, const F *Ax, const I *Ai, const I *Ap, const F *x, F *y) { /* common.c:598-639 */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (I j = 0; j < n; ++j) {
    F yj = y[j];
    for (I p = Ap[j]; p < Ap[j + 1]; ++p) yj += Ax[p] * x[Ai[p]];
    y[j] = yj;
  }
}
void orc_accum_by_A(I n, const F *Ax, const I *Ai, const I *Ap, const F *x, F *y) { /* common.c:644-695 (serial loop) */
  for (I j = 0; j < n; ++j) {
    F xj = x[j];
    for (I p = Ap[j]; p < Ap[j + 1]; ++p) y[Ai[p]] += Ax[p] * xj;
  }
}

/* ------------------------------------------------------------------------- */
/* scaling of A -- linsys/common.c:150-565                                    */
/* ------------------------------------------------------------------------- */
static F clamp_scale(F e, F lo, F hi) { if (e < lo) return 1; if (e > hi) return hi; return e; } /* e.g. common.c:224-229 */

void orc_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, F *D, F *E, F *mean_norm_row, F *mean_norm_col) {
  const I m = A->m, n = A->n;
  F *Dk = (F *)calloc((size_t)m, sizeof(F)), *Ek = (F *)calloc((size_t)n, sizeof(F));
  F *Dt = (F *)calloc((size_t)m, sizeof(F));
  F *D_pc = (F *)malloc(sizeof(F) * m), *E_pc = (F *)malloc(sizeof(F) * n);
  F *D_or = (F *)malloc(sizeof(F) * m), *E_or = (F *)malloc(sizeof(F) * n);
  F *D_rz = (F *)malloc(sizeof(F) * m), *E_rz = (F *)malloc(sizeof(F) * n);
  F *D_qp = (F *)malloc(sizeof(F) * m), *E_qp = (F *)malloc(sizeof(F) * n);
  const F min_row = MIN_SCALE * sqrt((F)n), max_row = MAX_SCALE * sqrt((F)n); /* common.c:172-175 */
  const F min_col = MIN_SCALE * sqrt((F)m), max_col = MAX_SCALE * sqrt((F)m);
  I i, j, k;

  for (i = 0; i < m; ++i) { D_pc[i] = D_or[i] = D_rz[i] = D_qp[i] = 1.0; }
  for (i = 0; i < n; ++i) { E_pc[i] = E_or[i] = E_rz[i] = E_qp[i] = 1.0; }

  if (stgs->pc_ruiz_rescale) { /* "pc" pass: sqrt of 1-norms, common.c:217-266 */
    memset(Dk, 0, sizeof(F) * m);
    for (i = 0; i < n; ++i) {
      I len = A->p[i + 1] - A->p[i];
      F e = clamp_scale(sqrt(v_nrm1(&A->x[A->p[i]], len)), min_col, max_col);
      v_scale(&A->x[A->p[i]], 1.0 / e, len);
      E_pc[i] = e;
    }
    for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) Dk[A->i[j]] += ABSF(A->x[j]);
    for (i = 0; i < m; ++i) D_pc[i] = clamp_scale(sqrt(Dk[i]), min_row, max_row);
    for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) A->x[j] /= D_pc[A->i[j]];
  }
  if (stgs->origin_rescale) { /* 2-norm pass, common.c:279-327 */
    memset(Dk, 0, sizeof(F) * m);
    for (i = 0; i < n; ++i) {
      I len = A->p[i + 1] - A->p[i];
      F e = clamp_scale(v_nrm2(&A->x[A->p[i]], len), min_col, max_col);
      v_scale(&A->x[A->p[i]], 1.0 / e, len);
      E_or[i] = e;
    }
    for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) Dk[A->i[j]] += A->x[j] * A->x[j];
    for (i = 0; i < m; ++i) D_or[i] = clamp_scale(sqrt(Dk[i]), min_row, max_row);
    for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) A->x[j] /= D_or[A->i[j]];
  }
  if (stgs->pc_ruiz_rescale) { /* Ruiz passes: sqrt of inf-norms, common.c:339-413 */
    for (k = 0; k < stgs->ruiz_iter; ++k) {
      memset(Dk, 0, sizeof(F) * m);
      for (i = 0; i < n; ++i) {
        I len = A->p[i + 1] - A->p[i];
        F e = clamp_scale(sqrt(v_nrminf(&A->x[A->p[i]], len)), min_col, max_col);
        v_scale(&A->x[A->p[i]], 1.0 / e, len);
        Ek[i] = e;
      }
      for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) { F w = ABSF(A->x[j]); if (w >= Dk[A->i[j]]) Dk[A->i[j]] = w; }
      for (i = 0; i < m; ++i) Dk[i] = clamp_scale(sqrt(Dk[i]), min_row, max_row);
      for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) A->x[j] /= Dk[A->i[j]];
      for (i = 0; i < n; ++i) E_rz[i] = E_rz[i] * Ek[i];
      for (i = 0; i < m; ++i) D_rz[i] = D_rz[i] * Dk[i];
    }
  }
  if (stgs->qp_rescale) { /* geometric-mean pass, common.c:415-499 */
    memset(D_qp, 0, sizeof(F) * m);
    for (i = 0; i < n; ++i) {
      I len = A->p[i + 1] - A->p[i];
      F e = v_nrminf(&A->x[A->p[i]], len);
      F t = v_minabs_sqrt(&A->x[A->p[i]], len, e);
      e = clamp_scale(t * sqrt(e), min_col, max_col);
      v_scale(&A->x[A->p[i]], 1.0 / e, len);
      E_qp[i] = e;
    }
    for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) { F w = ABSF(A->x[j]); if (w >= D_qp[A->i[j]]) D_qp[A->i[j]] = w; }
    for (i = 0; i < m; ++i) Dt[i] = D_qp[i];
    for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) { F w = ABSF(A->x[j]); if (w <= Dt[A->i[j]] && w > 0) Dt[A->i[j]] = w; }
    for (i = 0; i < m; ++i) D_qp[i] = clamp_scale(sqrt(D_qp[i] * Dt[i]), min_row, max_row);
    for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) A->x[j] /= D_qp[A->i[j]];
  }

  for (i = 0; i < m; ++i) D[i] = D_pc[i] * D_rz[i] * D_or[i] * D_qp[i]; /* common.c:512-520 */
  for (i = 0; i < n; ++i) E[i] = E_pc[i] * E_rz[i] * E_or[i] * E_qp[i];

  memset(Dk, 0, sizeof(F) * m); /* mean row / column 2-norms, common.c:523-545 */
  for (i = 0; i < n; ++i) for (j = A->p[i]; j < A->p[i + 1]; ++j) Dk[A->i[j]] += A->x[j] * A->x[j];
  *mean_norm_row = 0.0;
  for (i = 0; i < m; ++i) *mean_norm_row += sqrt(Dk[i]) / m;
  *mean_norm_col = 0.0;
  for (i = 0; i < n; ++i) *mean_norm_col += v_nrm2(&A->x[A->p[i]], A->p[i + 1] - A->p[i]) / n;

  if (stgs->scale != 1) v_scale(A->x, stgs->scale, A->p[n]); /* common.c:547-550 */

  free(Dk); free(Ek); free(Dt); free(D_pc); free(E_pc); free(D_or); free(E_or);
  free(D_rz); free(E_rz); free(D_qp); free(E_qp);
}

/* ------------------------------------------------------------------------- */
/* work struct                                                                */
/* ------------------------------------------------------------------------- */
typedef struct {
  /* indirect: explicit transpose + Jacobi-PCG scratch (indirect.h:14-29) */
  F *p, *r, *Gp, *tmp, *z, *M;
  F *Atx; I *Ati, *Atp; /* CSR of A (= CSC of A') */
  I tot_cg_its;
  /* direct: P, L (CSC, strictly lower), D, scratch (direct.h:16-27) */
  I N; I *P; I *Lp, *Li; F *Lx, *Dg, *bp;
} OrcLinSys;

struct ORC_WORK {
  I m, n;
  int linsys;
  ABIPMatrix A; /* own scaled copy */
  ABIPSettings *stgs;
  F sp;
  F sigma, gamma; I final_check, double_check;
  F mu, beta;
  F *u, *v, *u_t, *u_prev, *v_prev, *u_avg, *v_avg, *u_avgcon, *v_avgcon, *u_sumcon, *v_sumcon;
  I fre_old;
  F *h, *g, *pr, *dr, *b, *c;
  F g_th, sc_b, sc_c, nm_b, nm_c;
  F *D, *E; F mean_norm_row_A, mean_norm_col_A;
  OrcLinSys ls;
  /* adaptive (adaptive.c:13-32) */
  F *a_u_prev, *a_v_prev, *a_ut, *a_u, *a_v, *a_ut_next, *a_u_next, *a_v_next, *a_dut, *a_du, *a_dv;
  /* trace */
  I trace_T, trace_n; F *trace_buf;
};

typedef struct {
  I last_ipm_iter, last_admm_iter;
  F res_pri, res_dual, rel_gap, res_infeas, res_unbdd, ct_x_by_tau, bt_y_by_tau, tau, kap;
} OrcResid;

/* ------------------------------------------------------------------------- */
/* indirect back-end -- linsys/indirect.c                                     */
/* ------------------------------------------------------------------------- */
static void ind_init(OrcWork *w) {
  const ABIPMatrix *A = &w->A; OrcLinSys *p = &w->ls; const I m = A->m, n = A->n, nnz = A->p[n];
  p->p = (F *)malloc(sizeof(F) * m); p->r = (F *)malloc(sizeof(F) * m); p->Gp = (F *)malloc(sizeof(F) * m);
  p->tmp = (F *)malloc(sizeof(F) * n); p->z = (F *)malloc(sizeof(F) * m); p->M = (F *)malloc(sizeof(F) * m);
  p->Ati = (I *)malloc(sizeof(I) * (nnz > 0 ? nnz : 1)); p->Atx = (F *)malloc(sizeof(F) * (nnz > 0 ? nnz : 1));
  p->Atp = (I *)malloc(sizeof(I) * (m + 1));
  /* transpose, indirect.c:81-139: counting sort by row keeps columns ascending inside a row */
  I *cnt = (I *)calloc((size_t)m, sizeof(I));
  for (I q = 0; q < nnz; ++q) cnt[A->i[q]]++;
  I run = 0;
  for (I i = 0; i < m; ++i) { p->Atp[i] = run; run += cnt[i]; cnt[i] = p->Atp[i]; }
  p->Atp[m] = run;
  for (I j = 0; j < n; ++j)
    for (I q = A->p[j]; q < A->p[j + 1]; ++q) { I d = cnt[A->i[q]]++; p->Ati[d] = j; p->Atx[d] = A->x[q]; }
  free(cnt);
  /* Jacobi preconditioner, indirect.c:36-79: M_i = 1 / sum_j A_ij^2 (rho_y is NOT added) */
  memset(p->M, 0, sizeof(F) * m);
  for (I j = 0; j < n; ++j) for (I q = A->p[j]; q < A->p[j + 1]; ++q) p->M[A->i[q]] += A->x[q] * A->x[q];
  for (I i = 0; i < m; ++i) p->M[i] = 1 / p->M[i];
  p->tot_cg_its = 0;
}
static void ind_accum_A(OrcWork *w, const F *x, F *y) { /* indirect.c:233-242: gather on the stored transpose */
  orc_accum_by_Atrans(w->m, w->ls.Atx, w->ls.Ati, w->ls.Atp, x, y);
}
static void ind_accum_At(OrcWork *w, const F *x, F *y) { /* indirect.c:222-231 */
  orc_accum_by_Atrans(w->n, w->A.x, w->A.i, w->A.p, x, y);
}
static void ind_matvec(OrcWork *w, const F *x, F *y) { /* y = (rho_y I + A A') x, indirect.c:205-220 */
  memset(w->ls.tmp, 0, sizeof(F) * w->n);
  ind_accum_At(w, x, w->ls.tmp);
  memset(y, 0, sizeof(F) * w->m);
  ind_accum_A(w, w->ls.tmp, y);
  v_axpy(y, x, w->m, w->stgs->rho_y);
}
static void ind_precond(const F *M, F *z, const F *r, I m, F *ipzr) { /* indirect.c:263-280 */
  *ipzr = 0;
  for (I i = 0; i < m; ++i) { z[i] = r[i] * M[i]; *ipzr += z[i] * r[i]; }
}
static I ind_pcg(OrcWork *w, const F *s, F *b, I max_its, F tol) { /* indirect.c:321-391 */
  OrcLinSys *q = &w->ls; const I m = w->m; F ipzr, ipzr_old, alpha; I i;
  F *p = q->p, *Gp = q->Gp, *r = q->r, *z = q->z, *M = q->M;
  if (!s) { memcpy(r, b, sizeof(F) * m); memset(b, 0, sizeof(F) * m); }
  else { ind_matvec(w, s, r); v_axpy(r, b, m, -1); v_scale(r, -1, m); memcpy(b, s, sizeof(F) * m); }
  if (v_nrm2(r, m) < MINF(tol, 1e-18)) return 0;
  ind_precond(M, z, r, m, &ipzr);
  memcpy(p, z, sizeof(F) * m);
  for (i = 0; i < max_its; ++i) {
    ind_matvec(w, p, Gp);
    alpha = ipzr / v_dot(p, Gp, m);
    v_axpy(b, p, m, alpha);
    v_axpy(r, Gp, m, -alpha);
    if (v_nrm2(r, m) < tol) return i + 1;
    ipzr_old = ipzr;
    ind_precond(M, z, r, m, &ipzr);
    v_scale(p, ipzr / ipzr_old, m);
    v_axpy(p, z, m, 1);
  }
  return i;
}
static I ind_solve(OrcWork *w, F *b, const F *s, I iter) { /* indirect.c:393-434 */
  const I m = w->m, n = w->n;
  F cg_tol = v_nrm2(b, m) * (iter < 0 ? 1e-9 : 1e-1 / pow((F)iter + 1, w->stgs->cg_rate));
  cg_tol = MAXF(cg_tol, 1e-07);
  ind_accum_A(w, &b[m], b);
  I its = ind_pcg(w, s, b, m, MAXF(cg_tol, 1e-9));
  v_scale(&b[m], -1, n);
  ind_accum_At(w, b, &b[m]);
  if (iter >= 0) w->ls.tot_cg_its += its;
  return its;
}

/* ------------------------------------------------------------------------- */
/* direct back-end -- linsys/direct.c.  KKT assembly as direct.c:49-104; the   */
/* ordering (reference: SuiteSparse AMD, direct.c:106-119) is our own exact    */
/* minimum-degree on the quotient graph; the factorisation (reference:         */
/* LDL_symbolic/LDL_numeric, external/ldl/ldl.c) is our own up-looking LDL'.   */
/* The solve sequence perm / L / D / L' / perm' follows direct.c:172-198 and   */
/* external/ldl/ldl.c:357-550.                                                 */
/* ------------------------------------------------------------------------- */
static int dir_init(OrcWork *w) {
  const ABIPMatrix *A = &w->A; OrcLinSys *ls = &w->ls;
  const I m = A->m, n = A->n, N = m + n, nnzA = A->p[n];
  I i, j, q;
  ls->N = N;
  /* upper triangle of K = [[rho_y I, A],[A', -I]] in CSC (direct.c:49-104): column j+m holds A(:,j) then -1 */
  I *Kp = (I *)malloc(sizeof(I) * (N + 1)), *Ki = (I *)malloc(sizeof(I) * (N + nnzA)); F *Kx = (F *)malloc(sizeof(F) * (N + nnzA));
  I kk = 0;
  for (i = 0; i < m; ++i) { Kp[i] = kk; Ki[kk] = i; Kx[kk] = w->stgs->rho_y; ++kk; }
  for (j = 0; j < n; ++j) {
    Kp[m + j] = kk;
    for (q = A->p[j]; q < A->p[j + 1]; ++q) { Ki[kk] = A->i[q]; Kx[kk] = A->x[q]; ++kk; }
    Ki[kk] = m + j; Kx[kk] = -1; ++kk;
  }
  Kp[N] = kk;
  const int rc = orc_ldl_factor(N, Kp, Ki, Kx, &ls->P, &ls->Lp, &ls->Li, &ls->Lx, &ls->Dg);
  ls->bp = (F *)malloc(sizeof(F) * N);
  free(Kp); free(Ki); free(Kx);
  return rc;
}
static void dir_solve(OrcWork *w, F *b) { /* direct.c:172-198,305-328; ldl.c:357-550 */
  OrcLinSys *ls = &w->ls;
  orc_ldl_solve(ls->N, ls->P, ls->Lp, ls->Li, ls->Lx, ls->Dg, b, ls->bp);
}

/* dispatch: the reference's link-time variants (include/linsys.h) */
static void accum_by_A(OrcWork *w, const F *x, F *y) {
  if (w->linsys == ORC_LINSYS_INDIRECT) ind_accum_A(w, x, y);
  else orc_accum_by_A(w->n, w->A.x, w->A.i, w->A.p, x, y); /* direct.c:205-208: column scatter */
}
static void accum_by_Atrans(OrcWork *w, const F *x, F *y) { orc_accum_by_Atrans(w->n, w->A.x, w->A.i, w->A.p, x, y); }
static I solve_lin_sys(OrcWork *w, F *b, const F *s, I iter) {
  if (w->linsys == ORC_LINSYS_INDIRECT) return ind_solve(w, b, s, iter);
  dir_solve(w, b); return 0;
}
abip_int orc_lp_kkt_solve(OrcWork *w, F *rhs, const F *warm, I iter) { return solve_lin_sys(w, rhs, warm, iter); }

/* ------------------------------------------------------------------------- */
/* defaults / validation -- util.c:288-329, abip.c:1646-1734                  */
/* ------------------------------------------------------------------------- */
void orc_set_default_settings(ABIPData *d) {
  ABIPSettings *s = d->stgs;
  s->max_ipm_iters = 500; s->max_admm_iters = 1000000; s->eps = 1e-3; s->alpha = 1.8; s->cg_rate = 2.0;
  s->normalize = 1; s->scale = 1.0; s->rho_y = 1e-3; s->sparsity_ratio = 0.01;
  s->adaptive = 1; s->eps_cor = 0.2; s->eps_pen = 0.1; s->adaptive_lookback = 20;
  s->dynamic_x = 0.8; s->dynamic_eta = 1.1; s->restart_fre = 1000; s->restart_thresh = 100000;
  s->origin_rescale = 0; s->pc_ruiz_rescale = 1; s->qp_rescale = 0; s->ruiz_iter = 10;
  s->hybrid_mu = 1; s->dynamic_sigma = -1.0; s->hybrid_thresh = 1000; s->dynamic_sigma_second = 0.5;
  s->half_update = 0; s->avg_criterion = 0; s->verbose = 1; s->warm_start = 0;
}
static int validate(const ABIPData *d) {
  const ABIPSettings *s = d->stgs; const ABIPMatrix *A = d->A;
  if (d->m <= 0 || d->n <= 0) return -1;
  if (d->m > d->n) return -1;
  if (!A || !A->x || !A->i || !A->p) return -1; /* common.c:45-95 */
  for (I i = 0; i < A->n; ++i) if (A->p[i] > A->p[i + 1]) return -1;
  I nnz = A->p[A->n];
  if (((F)nnz / A->m > A->n) || nnz <= 0) return -1;
  for (I i = 0; i < nnz; ++i) if (A->i[i] > A->m - 1) return -1;
  if (s->max_ipm_iters <= 0 || s->max_admm_iters <= 0 || s->eps <= 0) return -1;
  if (s->alpha <= 0 || s->alpha >= 2 || s->rho_y <= 0 || s->scale <= 0) return -1;
  if (s->eps_cor <= 0 || s->eps_pen <= 0 || s->adaptive_lookback <= 0) return -1;
  if (s->hybrid_mu > 0 && s->dynamic_sigma >= 0) return -1;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* init / finish -- abip.c:1739-1839, 2301-2388                               */
/* ------------------------------------------------------------------------- */
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; }

void orc_lp_finish(OrcWork *w) {
  if (!w) return;
  free(w->A.x); free(w->A.i); free(w->A.p);
  free(w->u); free(w->v); free(w->u_t); free(w->u_prev); free(w->v_prev); free(w->u_avg); free(w->v_avg);
  free(w->u_avgcon); free(w->v_avgcon); free(w->u_sumcon); free(w->v_sumcon);
  free(w->h); free(w->g); free(w->pr); free(w->dr); free(w->b); free(w->c); free(w->D); free(w->E);
  OrcLinSys *p = &w->ls;
  free(p->p); free(p->r); free(p->Gp); free(p->tmp); free(p->z); free(p->M); free(p->Atx); free(p->Ati); free(p->Atp);
  free(p->P); free(p->Lp); free(p->Li); free(p->Lx); free(p->Dg); free(p->bp);
  free(w->a_u_prev); free(w->a_v_prev); free(w->a_ut); free(w->a_u); free(w->a_v); free(w->a_ut_next);
  free(w->a_u_next); free(w->a_v_next); free(w->a_dut); free(w->a_du); free(w->a_dv);
  free(w);
}

OrcWork *orc_lp_init(const ABIPData *d, ABIPInfo *info, int linsys) {
  if (!d || !info) return 0;
  if (validate(d) < 0) return 0;
  double t0 = now_ms();
  OrcWork *w = (OrcWork *)calloc(1, sizeof(OrcWork));
  const I m = d->m, n = d->n, l = m + n + 1, nnz = d->A->p[n];
  w->m = m; w->n = n; w->linsys = linsys; w->stgs = d->stgs; w->sp = d->sp;
  w->A.m = m; w->A.n = n;
  w->A.x = (F *)malloc(sizeof(F) * nnz); w->A.i = (I *)malloc(sizeof(I) * nnz); w->A.p = (I *)malloc(sizeof(I) * (n + 1));
  memcpy(w->A.x, d->A->x, sizeof(F) * nnz); memcpy(w->A.i, d->A->i, sizeof(I) * nnz); memcpy(w->A.p, d->A->p, sizeof(I) * (n + 1));
#define NEWV(len) ((F *)calloc((size_t)(len), sizeof(F)))
  w->u = NEWV(l); w->v = NEWV(l); w->u_t = NEWV(l); w->u_prev = NEWV(l); w->v_prev = NEWV(l);
  w->u_avg = NEWV(l); w->v_avg = NEWV(l); w->u_avgcon = NEWV(l); w->v_avgcon = NEWV(l); w->u_sumcon = NEWV(l); w->v_sumcon = NEWV(l);
  w->h = NEWV(l - 1); w->g = NEWV(l - 1); w->pr = NEWV(m); w->dr = NEWV(n); w->b = NEWV(m); w->c = NEWV(n);
  w->D = NEWV(m); w->E = NEWV(n);
  if (w->stgs->normalize) orc_normalize_A(&w->A, w->stgs, w->D, w->E, &w->mean_norm_row_A, &w->mean_norm_col_A);
  int rc = 0;
  if (linsys == ORC_LINSYS_INDIRECT) ind_init(w); else rc = dir_init(w);
  if (rc < 0) { orc_lp_finish(w); return 0; }
  if (w->stgs->adaptive_lookback > 0) { /* adaptive.c:258-303 */
    w->a_u_prev = NEWV(l); w->a_v_prev = NEWV(l); w->a_ut = NEWV(l); w->a_u = NEWV(l); w->a_v = NEWV(l);
    w->a_ut_next = NEWV(l); w->a_u_next = NEWV(l); w->a_v_next = NEWV(l); w->a_dut = NEWV(l); w->a_du = NEWV(l); w->a_dv = NEWV(l);
  }
#undef NEWV
  info->setup_time = now_ms() - t0;
  return w;
}

/* ------------------------------------------------------------------------- */
/* scaling of b, c and of the solution -- src/normalize.c                     */
/* ------------------------------------------------------------------------- */
static void normalize_b_c(OrcWork *w) { /* normalize.c:11-40 */
  I i; F nm;
  for (i = 0; i < w->n; ++i) w->c[i] /= w->E[i];
  nm = v_nrm2(w->c, w->n);
  w->sc_c = w->mean_norm_row_A / MAXF(nm, MIN_SCALE);
  for (i = 0; i < w->m; ++i) w->b[i] /= w->D[i];
  nm = v_nrm2(w->b, w->m);
  w->sc_b = w->mean_norm_col_A / MAXF(nm, MIN_SCALE);
  v_scale(w->c, w->sc_c * w->stgs->scale, w->n);
  v_scale(w->b, w->sc_b * w->stgs->scale, w->m);
}
static void un_normalize_sol(OrcWork *w, ABIPSolution *sol) { /* normalize.c:133-158 */
  I i;
  for (i = 0; i < w->n; ++i) sol->x[i] /= (w->E[i] * w->sc_b);
  for (i = 0; i < w->m; ++i) sol->y[i] /= (w->D[i] * w->sc_c);
  for (i = 0; i < w->n; ++i) sol->s[i] *= w->E[i] / (w->sc_c * w->stgs->scale);
}
static void normalize_warm_start(OrcWork *w) { /* normalize.c:101-128 */
  I i; F *y = w->u, *x = &w->u[w->m], *s = &w->v[w->m];
  for (i = 0; i < w->n; ++i) x[i] *= (w->E[i] * w->sc_b);
  for (i = 0; i < w->m; ++i) y[i] *= (w->D[i] * w->sc_c);
  for (i = 0; i < w->n; ++i) s[i] /= (w->E[i] / (w->sc_c * w->stgs->scale));
}

/* ------------------------------------------------------------------------- */
/* start points -- abip.c:307-381                                             */
/* ------------------------------------------------------------------------- */
static int is_nan(F x) { return x != x; }
static void warm_start_vars(OrcWork *w, const ABIPSolution *sol) { /* abip.c:307-357, quirks kept: the loop overwrites the warm start */
  const I n = w->n, m = w->m; I i;
  memset(w->v, 0, sizeof(F) * m);
  memcpy(w->u, sol->y, sizeof(F) * m);
  memcpy(&w->u[m], sol->x, sizeof(F) * n);
  memcpy(&w->v[m], sol->s, sizeof(F) * n);
  w->u[n + m] = 1.0; w->v[n + m] = 0.0;
  for (i = 0; i < n + m + 1; ++i) {
    if (is_nan(w->u[i]) && i < m) w->u[i] = 0; else w->u[i] = sqrt(w->mu / w->beta);
    if (is_nan(w->v[i])) w->v[i] = 0; else w->v[i] = sqrt(w->mu / w->beta);
  }
  if (w->stgs->normalize) normalize_warm_start(w);
}
static void cold_start_vars(OrcWork *w) { /* abip.c:361-381 */
  const I l = w->m + w->n + 1;
  memset(w->u, 0, sizeof(F) * w->m); memset(w->v, 0, sizeof(F) * w->m);
  for (I i = w->m; i < l; ++i) { w->u[i] = sqrt(w->mu / w->beta); w->v[i] = sqrt(w->mu / w->beta); }
}

/* ------------------------------------------------------------------------- */
/* residuals -- abip.c:385-535                                                */
/* ------------------------------------------------------------------------- */
static F calc_primal_resid(OrcWork *w, const F *x, F tau, F *nm_A_x) { /* abip.c:385-417 */
  F pres = 0, scale, *pr = w->pr; *nm_A_x = 0;
  memset(pr, 0, sizeof(F) * w->m);
  accum_by_A(w, x, pr);
  for (I i = 0; i < w->m; ++i) {
    scale = w->stgs->normalize ? w->D[i] / (w->sc_b * w->stgs->scale) : 1;
    scale = scale * scale;
    *nm_A_x += (pr[i] * pr[i]) * scale;
    pres += (pr[i] - w->b[i] * tau) * (pr[i] - w->b[i] * tau) * scale;
  }
  *nm_A_x = sqrt(*nm_A_x);
  return sqrt(pres);
}
static F calc_dual_resid(OrcWork *w, const F *y, const F *s, F tau, F *nm_At_ys) { /* abip.c:421-453 */
  F dres = 0, scale, *dr = w->dr; *nm_At_ys = 0;
  memset(dr, 0, sizeof(F) * w->n);
  accum_by_Atrans(w, y, dr);
  v_axpy(dr, s, w->n, 1.0);
  for (I i = 0; i < w->n; ++i) {
    scale = w->stgs->normalize ? w->E[i] / (w->sc_c * w->stgs->scale) : 1;
    scale = scale * scale;
    *nm_At_ys += (dr[i] * dr[i]) * scale;
    dres += (dr[i] - w->c[i] * tau) * (dr[i] - w->c[i] * tau) * scale;
  }
  *nm_At_ys = sqrt(*nm_At_ys);
  return sqrt(dres);
}
static void calc_residuals(OrcWork *w, OrcResid *r, I ipm_iter, I admm_iter) { /* abip.c:458-535 */
  const I n = w->n, m = w->m;
  const F *uu = w->stgs->avg_criterion ? w->u_avgcon : w->u, *vv = w->stgs->avg_criterion ? w->v_avgcon : w->v;
  const F *y = uu, *x = &uu[m], *s = &vv[m];
  F nmpr_tau, nmdr_tau, nm_A_x_tau, nm_At_ys_tau, ct_x, bt_y;
  const F den = w->stgs->normalize ? (w->stgs->scale * w->sc_c * w->sc_b) : 1;
  if (admm_iter && r->last_admm_iter == admm_iter) return;
  r->last_ipm_iter = ipm_iter; r->last_admm_iter = admm_iter;
  r->tau = ABSF(uu[n + m]);
  r->kap = ABSF(vv[n + m]) / den;
  nmpr_tau = calc_primal_resid(w, x, r->tau, &nm_A_x_tau);
  nmdr_tau = calc_dual_resid(w, y, s, r->tau, &nm_At_ys_tau);
  r->bt_y_by_tau = v_dot(y, w->b, m) / den;
  r->ct_x_by_tau = v_dot(x, w->c, n) / den;
  r->res_infeas = r->bt_y_by_tau > 0 ? w->nm_b * nm_At_ys_tau / r->bt_y_by_tau : NAN;
  r->res_unbdd = r->ct_x_by_tau < 0 ? w->nm_c * nm_A_x_tau / -r->ct_x_by_tau : NAN;
  bt_y = SAFEDIV_POS(r->bt_y_by_tau, r->tau);
  ct_x = SAFEDIV_POS(r->ct_x_by_tau, r->tau);
  r->res_pri = SAFEDIV_POS(nmpr_tau / (1 + w->nm_b), r->tau);
  r->res_dual = SAFEDIV_POS(nmdr_tau / (1 + w->nm_c), r->tau);
  r->rel_gap = ABSF(ct_x - bt_y) / (1 + ABSF(ct_x) + ABSF(bt_y));
}

/* ------------------------------------------------------------------------- */
/* the ADMM step pieces -- abip.c:539-748                                     */
/* ------------------------------------------------------------------------- */
/* rhs build + KKT solve + tau recovery; shared by project_lin_sys (abip.c:539-562)
 * and the two look-ahead steps of adaptive.c:92-99,124-131 */
static I lin_projection(OrcWork *w, F *ut, const F *u, const F *v, const F *warm, I iter) {
  const I n = w->n, m = w->m, l = n + m + 1;
  memcpy(ut, u, sizeof(F) * l);
  v_axpy(ut, v, l, 1.0);
  v_scale(ut, w->stgs->rho_y, m);
  v_axpy(ut, w->h, l - 1, -ut[l - 1]);
  v_axpy(ut, w->h, l - 1, -v_dot(ut, w->g, l - 1) / (w->g_th + 1));
  v_scale(&ut[m], -1, n);
  solve_lin_sys(w, ut, warm, iter);
  ut[l - 1] += v_dot(ut, w->h, l - 1);
  return 0;
}
static void project_barrier(OrcWork *w) { /* abip.c:717-748 */
  const I m = w->m, l = m + w->n + 1; I i; F tmp;
  for (i = 0; i < m; ++i) w->u[i] = w->u_t[i] - w->v[i];
  for (i = m; i < l; ++i) w->u[i] = w->stgs->alpha * w->u_t[i] + (1 - w->stgs->alpha) * w->u_prev[i] - w->v[i];
  for (i = m; i < l; ++i) { tmp = w->u[i] / 2; w->u[i] = tmp + sqrt(tmp * tmp + w->mu / w->beta); }
}
static void update_dual_vars(OrcWork *w) { /* abip.c:567-584 */
  const I m = w->m, l = m + w->n + 1;
  for (I i = m; i < l; ++i) w->v[i] += (w->u[i] - w->stgs->alpha * w->u_t[i] - (1.0 - w->stgs->alpha) * w->u_prev[i]);
}
static void half_update_dual_vars(OrcWork *w) { /* abip.c:663-679 */
  const I l = w->m + w->n + 1;
  for (I i = 0; i < l; ++i) w->v[i] += 0.5 * (w->u[i] - w->u_t[i]);
}
static void project_barrier_dual(OrcWork *w) { /* abip.c:681-711 */
  const I m = w->m, l = m + w->n + 1; I i; F tmp;
  for (i = 0; i < l; ++i) w->u[i] = w->u_t[i] - w->v[i];
  for (i = m; i < l; ++i) { tmp = w->u[i] / 2; w->u[i] = tmp + sqrt(tmp * tmp + w->mu / w->beta); }
  for (i = 0; i < l; ++i) w->v[i] += (w->u[i] - w->u_t[i]);
}
static void restart_vars(OrcWork *w, I admm_iter, I total_admm_iter) { /* abip.c:587-630 */
  const I fre = w->stgs->restart_fre, l = w->m + w->n + 1; I i;
  for (i = 0; i < l; ++i) { w->u_avg[i] += w->u[i]; w->v_avg[i] += w->v[i]; }
  if (total_admm_iter < w->stgs->restart_thresh || (admm_iter + 1 - w->fre_old) % fre != 0) return;
  for (i = 0; i < l; ++i) { w->u_avg[i] /= fre; w->v_avg[i] /= fre; }
  memcpy(w->u, w->u_avg, sizeof(F) * l); memcpy(w->v, w->v_avg, sizeof(F) * l);
  memset(w->u_avg, 0, sizeof(F) * l); memset(w->v_avg, 0, sizeof(F) * l);
  w->fre_old = fre;
}
static void compute_avg(OrcWork *w, I admm_iter) { /* abip.c:635-659 */
  const I l = w->m + w->n + 1, dom = admm_iter + 1;
  for (I i = 0; i < l; ++i) {
    w->u_sumcon[i] += w->u[i]; w->v_sumcon[i] += w->v[i];
    w->u_avgcon[i] = w->u_sumcon[i] / dom; w->v_avgcon[i] = w->v_sumcon[i] / dom;
  }
}

/* inner stopping metric -- abip.c:1951-2051 */
static F q_norm_of(OrcWork *w, const F *uu, const F *vv, F *pr, F *dr, F *norm_out) {
  const I m = w->m, n = w->n, l = m + n + 1; I i;
  const F *y = uu, *x = &uu[m], *s = &vv[m]; const F tau = uu[m + n], kap = vv[m + n];
  F Q = 0;
  memset(pr, 0, sizeof(F) * m); memset(dr, 0, sizeof(F) * n);
  accum_by_A(w, x, pr);
  accum_by_Atrans(w, y, dr);
  v_axpy(dr, s, n, 1.0);
  for (i = 0; i < m; ++i) Q += (pr[i] - w->b[i] * tau) * (pr[i] - w->b[i] * tau);
  for (i = 0; i < n; ++i) Q += (dr[i] - w->c[i] * tau) * (dr[i] - w->c[i] * tau);
  F cTx = v_dot(x, w->c, n), bTy = v_dot(y, w->b, m);
  Q += (bTy - cTx - kap) * (bTy - cTx - kap);
  *norm_out = 1 + sqrt(v_nrm2sq(uu, l) + v_nrm2sq(vv, l));
  return Q;
}
static F iterate_Q_norm_resd(OrcWork *w, I j) {
  F norm, Qres = q_norm_of(w, w->u, w->v, w->pr, w->dr, &norm);
  F Qres_avg = (F)w->stgs->max_admm_iters, norm_avg = 1;
  if ((j + 1) % 10 == 0) {
    F *pr_avg = (F *)malloc(sizeof(F) * w->m), *dr_avg = (F *)malloc(sizeof(F) * w->n);
    Qres_avg = q_norm_of(w, w->u_avgcon, w->v_avgcon, pr_avg, dr_avg, &norm_avg);
    free(pr_avg); free(dr_avg);
  }
  if (sqrt(Qres_avg) / norm_avg < sqrt(Qres) / norm) { w->stgs->avg_criterion = 1; return sqrt(Qres_avg) / norm_avg; }
  w->stgs->avg_criterion = 0;
  return sqrt(Qres) / norm;
}

/* ------------------------------------------------------------------------- */
/* barrier-parameter strategies -- abip.c:753-992                             */
/* ------------------------------------------------------------------------- */
static F gamma_table(F ratio, F top) { /* the shared ladder of abip.c:766-801 / 833-868 */
  if (ratio > 10.0) return top;
  if (ratio > 1.0) return 1.0;
  if (ratio > 0.5) return 0.9;
  if (ratio > 0.1) return 0.8;
  if (ratio > 0.05) return 0.7;
  if (ratio > 0.01) return 0.6;
  if (ratio > 0.005) return 0.5;
  if (ratio > 0.001) return 0.4;
  return 0.3;
}
static void update_barrier(OrcWork *w, const OrcResid *r) { /* abip.c:753-921 */
  F sigma, gamma, mu = w->mu;
  const F ratio = w->mu / w->stgs->eps;
  const F err_ratio = MAXF(MAXF(r->res_pri, r->res_dual), r->rel_gap) / w->stgs->eps;
  if (MAXF(w->sp, w->stgs->sparsity_ratio) > 0.4 || MINF(w->sp, w->stgs->sparsity_ratio) > 0.1) {
    gamma = gamma_table(ratio, 2.0);
    if (err_ratio > 6 && err_ratio <= 10) sigma = 0.5;
    else if (err_ratio > 3 && err_ratio <= 6) { sigma = 0.6; gamma = gamma * 0.8; }
    else if (err_ratio > 1 && err_ratio <= 3) { w->final_check = 1; gamma = gamma * 0.4; sigma = (ratio < 0.1) ? 0.8 : 0.7; }
    else sigma = w->sigma;
  } else {
    gamma = gamma_table(ratio, 3.0);
    if (err_ratio > 6 && err_ratio <= 10) { sigma = 0.82; gamma = gamma * 0.8; }
    else if (err_ratio > 4 && err_ratio <= 6) { sigma = 0.84; gamma = gamma * 0.6; }
    else if (err_ratio > 3 && err_ratio <= 4) { sigma = 0.85; gamma = gamma * 0.5; w->final_check = 1; }
    else if (err_ratio > 1 && err_ratio <= 3) {
      w->final_check = 1;
      if (ratio < 0.1) {
        if (w->double_check) { sigma = 0.9; gamma = gamma * 0.4; w->double_check = 0; }
        else { sigma = 1.0; gamma = gamma * 0.1; w->double_check = 1; }
      } else { sigma = 0.88; gamma = gamma * 0.4; }
    } else sigma = w->sigma;
  }
  w->mu = mu * sigma; w->sigma = sigma; w->gamma = gamma;
}
static int update_barrier_dynamic(OrcWork *w) { /* LOQO rule, abip.c:930-977 */
  const I m = w->m, n = w->n, l = m + n + 1;
  const F *u = w->stgs->avg_criterion ? w->u_avgcon : w->u, *v = w->stgs->avg_criterion ? w->v_avgcon : w->v;
  const double shrink = w->stgs->dynamic_sigma;
  double ksi, sigma, xisi, xs = 0.0, minxs = 1e+10;
  for (I i = m; i < l; ++i) { xisi = u[i] * v[i]; xs += xisi; minxs = MINF(xisi, minxs); }
  if (minxs <= 0.0) return -1; /* the reference asserts here (abip.c:967-970) */
  xs /= (n + 1); ksi = minxs / xs;
  sigma = MINF(0.05 * (1 - ksi) / ksi, 2.0);
  sigma = MAXF(0.1 * sigma * sigma * sigma, shrink);
  w->mu *= sigma;
  return 0;
}
static void update_barrier_dynamic_2(OrcWork *w) { /* abip.c:982-992: reads dynamic_sigma as the exponent */
  const F x = w->stgs->dynamic_x, eta = w->stgs->dynamic_sigma;
  w->mu *= MINF(x * w->mu, pow(w->mu, eta));
}
static void reinitialize_vars(OrcWork *w, I indx) { /* abip.c:996-1075 */
  const I m = w->m, l = m + w->n + 1; I i;
  F *u = w->stgs->avg_criterion ? w->u_avgcon : w->u, *v = w->stgs->avg_criterion ? w->v_avgcon : w->v;
  if (indx == 0) { for (i = m; i < l; ++i) { if (u[i] > v[i]) v[i] = w->sigma * v[i]; else u[i] = w->sigma * u[i]; } }
  else if (indx == 1) { for (i = m; i < l; ++i) { u[i] = sqrt(w->sigma) * u[i]; v[i] = sqrt(w->sigma) * v[i]; } }
  else { for (i = m; i < l; ++i) { u[i] = sqrt(1.0 / w->sigma) * u[i]; v[i] = sqrt(1.0 / w->sigma) * v[i]; } }
}

/* ------------------------------------------------------------------------- */
/* Barzilai-Borwein penalty search -- adaptive.c:34-256                       */
/* ------------------------------------------------------------------------- */
static I update_adapt_params(OrcWork *w, I iter) {
  F *u_prev = w->a_u_prev, *v_prev = w->a_v_prev, *ut = w->a_ut, *u = w->a_u, *v = w->a_v;
  F *ut_next = w->a_ut_next, *u_next = w->a_u_next, *v_next = w->a_v_next, *dut = w->a_dut, *du = w->a_du, *dv = w->a_dv;
  const I n = w->n, m = w->m, l = n + m + 1, K = w->stgs->adaptive_lookback; I i, j;
  const F al = w->stgs->alpha; F tmp, beta_prev = 1.0, beta = 0.0;
  memcpy(u_prev, w->u, sizeof(F) * l); memcpy(v_prev, w->v, sizeof(F) * l);
  for (i = 0; i < K; ++i) {
    lin_projection(w, ut, u_prev, v_prev, u_prev, iter);
    for (j = 0; j < m; ++j) u[j] = ut[j] - v_prev[j];
    for (j = m; j < l; ++j) u[j] = al * ut[j] + (1 - al) * u_prev[j] - v_prev[j];
    for (j = m; j < l; ++j) { tmp = u[j] / 2; u[j] = tmp + sqrt(tmp * tmp + w->mu / beta_prev); }
    for (j = m; j < l; ++j) v[j] = v_prev[j] + (u[j] - al * ut[j] - (1 - al) * u_prev[j]);
    /* NB: v[0:m) is never written here: it keeps whatever the previous pass (or calloc) left (adaptive.c:118-121) */
    lin_projection(w, ut_next, u, v, u, iter);
    for (j = 0; j < m; ++j) u_next[j] = ut_next[j] - v[j];
    for (j = m; j < l; ++j) u_next[j] = al * ut_next[j] + (1 - al) * u[j] - v[j];
    for (j = m; j < l; ++j) { tmp = u_next[j] / 2; u_next[j] = tmp + sqrt(tmp * tmp + w->mu / beta_prev); }
    for (j = m; j < l; ++j) v_next[j] = v[j] + (u_next[j] - al * ut_next[j] - (1 - al) * u[j]);

    memcpy(dut, v, sizeof(F) * l); v_scale(dut, 2.0, l); v_axpy(dut, u_next, l, 1.0); v_axpy(dut, u, l, -1.0);
    v_axpy(dut, v_next, l, -1.0); v_axpy(dut, v_prev, l, -1.0);
    memcpy(du, u, sizeof(F) * l); v_axpy(du, u_next, l, -1.0);
    memcpy(dv, u_next, sizeof(F) * l); v_axpy(dv, u, l, -1.0); v_scale(dv, al - 1.0, l); v_axpy(dv, v_next, l, 1.0); v_axpy(dv, v, l, -1.0);

    F utut = v_dot(dut, dut, l), utv = v_dot(dut, dv, l), uu = v_dot(du, du, l), vv = v_dot(dv, dv, l), uv = v_dot(du, dv, l);
    F norm_ut = v_nrm2(dut, l), norm_u = v_nrm2(du, l), norm_v = v_nrm2(dv, l);
    F alpha_SD = vv / utv, alpha_MG = utv / utut, gamma_SD = vv / uv, gamma_MG = uv / uu;
    F alpha_ss = (2 * alpha_MG > alpha_SD) ? alpha_MG : alpha_SD - 0.5 * alpha_MG;
    F gamma_ss = (2 * gamma_MG > gamma_SD) ? gamma_MG : gamma_SD - 0.5 * gamma_MG;
    F alpha_cor = utv / (norm_v * norm_ut), gamma_cor = uv / (norm_v * norm_u);
    const F ec = w->stgs->eps_cor;
    if (alpha_cor > ec && gamma_cor > ec) beta = sqrt(alpha_ss * gamma_ss);
    else if (alpha_cor > ec && gamma_cor <= ec) beta = alpha_ss;
    else if (alpha_cor <= ec && gamma_cor > ec) beta = gamma_ss;
    else beta = beta_prev;

    if (ABSF(beta - beta_prev) > 0 && ABSF(beta - beta_prev) <= w->stgs->eps_pen) { beta = (beta + beta_prev) / 2; break; }
    else if (ABSF(beta - beta_prev) > w->stgs->eps_pen) {
      beta_prev = beta;
      memcpy(u_prev, u, sizeof(F) * l);
      for (j = 0; j < m; ++j) v_prev[j] = v[j];
      for (j = m; j < l; ++j) v_prev[j] = (w->mu / beta_prev) / u_prev[j];
    } else { memcpy(u_prev, u, sizeof(F) * l); memcpy(v_prev, v, sizeof(F) * l); }
  }
  w->beta = beta;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* status / solution extraction -- abip.c:1100-1414, 1613-1641                */
/* ------------------------------------------------------------------------- */
static I has_converged(OrcWork *w, const OrcResid *r, I ipm_iter, I admm_iter) { /* abip.c:1613-1641 */
  const F eps = w->stgs->eps;
  if (r->res_pri < eps && (r->res_dual < eps || w->stgs->pfeasopt) && r->rel_gap < eps) return ABIP_SOLVED;
  if (r->res_unbdd < eps && ipm_iter > 0 && admm_iter > 0) return ABIP_UNBOUNDED;
  if (r->res_infeas < eps && ipm_iter > 0 && admm_iter > 0) return ABIP_INFEASIBLE;
  return 0;
}
static int st_solved(I s) { return s == ABIP_SOLVED || s == ABIP_SOLVED_INACCURATE; }
static int st_infeas(I s) { return s == ABIP_INFEASIBLE || s == ABIP_INFEASIBLE_INACCURATE; }
static int st_unbdd(I s) { return s == ABIP_UNBOUNDED || s == ABIP_UNBOUNDED_INACCURATE; }

static void get_solution(OrcWork *w, ABIPSolution *sol, ABIPInfo *info, OrcResid *r, I ipm_iter, I admm_iter) { /* abip.c:1344-1414 */
  const I m = w->m, n = w->n, l = m + n + 1;
  calc_residuals(w, r, ipm_iter, admm_iter);
  const F *uu = w->stgs->avg_criterion ? w->u_avgcon : w->u, *vv = w->stgs->avg_criterion ? w->v_avgcon : w->v;
  if (!sol->x) sol->x = (F *)malloc(sizeof(F) * n);
  if (!sol->y) sol->y = (F *)malloc(sizeof(F) * m);
  if (!sol->s) sol->s = (F *)malloc(sizeof(F) * n);
  memcpy(sol->x, &uu[m], sizeof(F) * n); memcpy(sol->y, uu, sizeof(F) * m); memcpy(sol->s, &vv[m], sizeof(F) * n);
  int kind; /* 0 solved, 1 indeterminate, 2 infeasible, 3 unbounded */
  if (info->status_val == ABIP_UNFINISHED) {
    if (r->tau > INDETERMINATE_TOL && r->tau > r->kap) kind = 0;
    else if (v_nrm2(uu, l) < INDETERMINATE_TOL * sqrt((F)l)) kind = 1;
    else if (-r->bt_y_by_tau < r->ct_x_by_tau) kind = 2;
    else kind = 3;
  } else if (st_solved(info->status_val)) kind = 0;
  else if (st_infeas(info->status_val)) kind = 2;
  else kind = 3;
  const int inacc = (info->status_val == 0);
  if (kind == 0) { /* abip.c:1100-1123 */
    F sc = SAFEDIV_POS(1.0, r->tau);
    v_scale(sol->x, sc, n); v_scale(sol->y, sc, m); v_scale(sol->s, sc, n);
    if (inacc) { strcpy(info->status, "Solved/Inaccurate"); info->status_val = ABIP_SOLVED_INACCURATE; }
    else { strcpy(info->status, "Solved"); info->status_val = ABIP_SOLVED; }
  } else if (kind == 1) { /* abip.c:1079-1095 */
    strcpy(info->status, "Indeterminate");
    v_scale(sol->x, NAN, n); v_scale(sol->y, NAN, m); v_scale(sol->s, NAN, n);
    info->status_val = ABIP_INDETERMINATE;
  } else if (kind == 2) { /* abip.c:1214-1236 */
    v_scale(sol->y, 1 / r->bt_y_by_tau, m); v_scale(sol->s, 1 / r->bt_y_by_tau, n); v_scale(sol->x, NAN, n);
    if (inacc) { strcpy(info->status, "Infeasible/Inaccurate"); info->status_val = ABIP_INFEASIBLE_INACCURATE; }
    else { strcpy(info->status, "Infeasible"); info->status_val = ABIP_INFEASIBLE; }
  } else { /* abip.c:1240-1262 */
    v_scale(sol->x, -1 / r->ct_x_by_tau, n); v_scale(sol->y, NAN, m); v_scale(sol->s, NAN, n);
    if (inacc) { strcpy(info->status, "Unbounded/Inaccurate"); info->status_val = ABIP_UNBOUNDED_INACCURATE; }
    else { strcpy(info->status, "Unbounded"); info->status_val = ABIP_UNBOUNDED; }
  }
  if (w->stgs->normalize) un_normalize_sol(w, sol);
  /* get_info, abip.c:1296-1340 */
  info->ipm_iter = ipm_iter + 1; info->admm_iter = admm_iter + 1;
  info->res_infeas = r->res_infeas; info->res_unbdd = r->res_unbdd;
  if (st_solved(info->status_val)) {
    info->rel_gap = r->rel_gap; info->res_pri = r->res_pri; info->res_dual = r->res_dual;
    info->pobj = r->ct_x_by_tau / r->tau; info->dobj = r->bt_y_by_tau / r->tau;
  } else if (st_unbdd(info->status_val)) {
    info->rel_gap = NAN; info->res_pri = NAN; info->res_dual = NAN; info->pobj = -INFINITY; info->dobj = -INFINITY;
  } else if (st_infeas(info->status_val)) {
    info->rel_gap = NAN; info->res_pri = NAN; info->res_dual = NAN; info->pobj = INFINITY; info->dobj = INFINITY;
  }
}

/* ------------------------------------------------------------------------- */
/* update_work -- abip.c:1843-1927                                            */
/* ------------------------------------------------------------------------- */
static void update_work(const ABIPData *d, OrcWork *w, const ABIPSolution *sol) {
  const I n = d->n, m = d->m;
  w->nm_b = v_nrm2(d->b, m); w->nm_c = v_nrm2(d->c, n);
  memcpy(w->b, d->b, sizeof(F) * m); memcpy(w->c, d->c, sizeof(F) * n);
  if (w->stgs->normalize) normalize_b_c(w);
  const F mx = MAXF(w->sp, w->stgs->sparsity_ratio), mn = MINF(w->sp, w->stgs->sparsity_ratio);
  if (mx > 0.4 || (mn > 0.1 && mn < 0.2)) { w->sigma = 0.3; w->gamma = 2.0; }
  else if (mn > 0.2) { w->sigma = 0.5; w->gamma = 3.0; }
  else { w->sigma = 0.8; w->gamma = 3.0; }
  w->final_check = 0; w->double_check = 0; w->mu = 1.0; w->beta = 1.0;
  if (w->stgs->warm_start) warm_start_vars(w, sol); else cold_start_vars(w);
  memcpy(w->h, w->b, sizeof(F) * m); memcpy(&w->h[m], w->c, sizeof(F) * n);
  v_scale(w->h, -1, m);
  memcpy(w->g, w->h, sizeof(F) * (n + m));
  solve_lin_sys(w, w->g, 0, -1);
  v_scale(&w->g[m], -1, n);
  w->g_th = v_dot(w->h, w->g, n + m);
}

static void trace_push(OrcWork *w) {
  if (!w->trace_buf || w->trace_n >= w->trace_T) return;
  const I l = w->m + w->n + 1; F *dst = w->trace_buf + 3 * l * w->trace_n;
  memcpy(dst, w->u, sizeof(F) * l); memcpy(dst + l, w->v, sizeof(F) * l); memcpy(dst + 2 * l, w->u_t, sizeof(F) * l);
  w->trace_n++;
}

/* ------------------------------------------------------------------------- */
/* the solve loop -- abip.c:2056-2297                                         */
/* ------------------------------------------------------------------------- */
abip_int orc_lp_solve(OrcWork *w, const ABIPData *d, ABIPSolution *sol, ABIPInfo *info) {
  if (!d || !sol || !info || !w || !d->b || !d->c) return ABIP_FAILED;
  I i, j, k = 0, inner_stopper; const I l = w->m + w->n + 1;
  OrcResid r; memset(&r, 0, sizeof(r));
  clock_t start_time = clock();
  const double maxTime = w->stgs->max_time;
  const double t0 = now_ms();
  info->status_val = ABIP_UNFINISHED;
  r.last_ipm_iter = -1; r.last_admm_iter = -1;
  update_work(d, w, sol);

  for (i = 0; i < w->stgs->max_ipm_iters; ++i) {
    const F mn = MINF(w->sp, w->stgs->sparsity_ratio);
    if (mn > 0.5) inner_stopper = (int)round(pow(w->mu, -0.35));
    else if (mn > 0.2) inner_stopper = (int)round(pow(w->mu, -1));
    else inner_stopper = w->stgs->max_admm_iters;
    w->fre_old = 0;
    memset(w->u_avg, 0, sizeof(F) * l); memset(w->v_avg, 0, sizeof(F) * l);
    memset(w->u_sumcon, 0, sizeof(F) * l); memset(w->v_sumcon, 0, sizeof(F) * l);
    if (w->stgs->avg_criterion) { memcpy(w->u, w->u_avgcon, sizeof(F) * l); memcpy(w->v, w->v_avgcon, sizeof(F) * l); }

    for (j = 0; j < inner_stopper; ++j) {
      memcpy(w->u_prev, w->u, sizeof(F) * l); memcpy(w->v_prev, w->v, sizeof(F) * l);
      lin_projection(w, w->u_t, w->u, w->v, w->u, k); /* project_lin_sys, abip.c:539-562 */
      if (w->stgs->half_update) { half_update_dual_vars(w); project_barrier_dual(w); }
      else { project_barrier(w); update_dual_vars(w); }
      restart_vars(w, j, k);
      compute_avg(w, j);
      trace_push(w);
      k += 1;
      if (iterate_Q_norm_resd(w, j) < w->gamma * w->mu) {
        if (w->stgs->half_update) for (I jj = 0; jj < l; ++jj) if (w->v[jj] < 0) w->v[jj] = 1e-6;
        break;
      }
      if (w->final_check) {
        calc_residuals(w, &r, i, k);
        if ((info->status_val = has_converged(w, &r, i, k)) != 0 || k + 1 >= w->stgs->max_admm_iters || i + 1 >= w->stgs->max_ipm_iters) {
          get_solution(w, sol, info, &r, i, k);
          info->solve_time = now_ms() - t0;
          return info->status_val;
        }
      }
    }
    double elapsedT = ((F)clock() - start_time) / CLOCKS_PER_SEC;
    if (elapsedT > maxTime) w->stgs->max_admm_iters = k * 1.05;
    if (w->mu < w->stgs->eps) w->final_check = 1;
    calc_residuals(w, &r, i, k);
    if ((info->status_val = has_converged(w, &r, i, k)) != 0 || k + 1 >= w->stgs->max_admm_iters) {
      get_solution(w, sol, info, &r, i, k);
      info->solve_time = now_ms() - t0;
      return info->status_val;
    }
    if (w->stgs->hybrid_mu) { /* abip.c:2251-2277 */
      if (w->stgs->dynamic_sigma_second > 0.0 && w->mu < w->stgs->hybrid_thresh * w->stgs->eps) {
        w->stgs->dynamic_sigma = w->stgs->dynamic_sigma_second;
        if (update_barrier_dynamic(w) < 0) { strcpy(info->status, "Failure"); info->status_val = ABIP_FAILED; return ABIP_FAILED; }
      } else if (w->stgs->dynamic_sigma_second == 0.0 && w->mu < w->stgs->hybrid_thresh * w->stgs->eps) {
        w->stgs->dynamic_sigma = w->stgs->dynamic_sigma_second;
        update_barrier(w, &r);
      } else if (w->stgs->dynamic_sigma < 0.0) update_barrier_dynamic_2(w);
    } else {
      if (w->stgs->dynamic_sigma == 0.0) update_barrier(w, &r);
      else if (w->stgs->dynamic_sigma < 0.0) update_barrier_dynamic_2(w);
      else if (update_barrier_dynamic(w) < 0) { strcpy(info->status, "Failure"); info->status_val = ABIP_FAILED; return ABIP_FAILED; }
    }
    reinitialize_vars(w, 0);
    if (w->stgs->adaptive) {
      reinitialize_vars(w, 1);
      w->beta = 1;
      update_adapt_params(w, k); /* ABIP(adaptive), adaptive.c:305-334 */
      reinitialize_vars(w, 2);
    }
  }
  return info->status_val;
}

abip_int orc_lp_main(const ABIPData *d, ABIPSolution *sol, ABIPInfo *info, int linsys) { /* abip.c:2393-2422 */
  OrcWork *w = orc_lp_init(d, info, linsys);
  I status;
  if (w) { orc_lp_solve(w, d, sol, info); status = info->status_val; }
  else {
    status = ABIP_FAILED;
    if (info) { info->status_val = status; strcpy(info->status, "Failure"); info->ipm_iter = -1; info->admm_iter = -1; }
  }
  orc_lp_finish(w);
  return status;
}

/* ------------------------------------------------------------------------- */
/* introspection                                                              */
/* ------------------------------------------------------------------------- */
const abip_float *orc_lp_vec(const OrcWork *w, const char *name, abip_int *len) {
  const I m = w->m, n = w->n, l = m + n + 1;
#define RET(nm, ptr, ln) if (!strcmp(name, nm)) { if (len) *len = (ln); return (ptr); }
  RET("u", w->u, l) RET("v", w->v, l) RET("u_t", w->u_t, l) RET("u_prev", w->u_prev, l)
  RET("u_avgcon", w->u_avgcon, l) RET("v_avgcon", w->v_avgcon, l)
  RET("h", w->h, l - 1) RET("g", w->g, l - 1) RET("b", w->b, m) RET("c", w->c, n)
  RET("D", w->D, m) RET("E", w->E, n) RET("Ax", w->A.x, w->A.p[n])
#undef RET
  return 0;
}
abip_float orc_lp_scalar(const OrcWork *w, const char *name) {
#define RET(nm, val) if (!strcmp(name, nm)) return (F)(val);
  RET("mu", w->mu) RET("beta", w->beta) RET("sigma", w->sigma) RET("gamma", w->gamma) RET("g_th", w->g_th)
  RET("sc_b", w->sc_b) RET("sc_c", w->sc_c) RET("nm_b", w->nm_b) RET("nm_c", w->nm_c)
  RET("tot_cg_its", w->ls.tot_cg_its)
  RET("lnnz", (w->linsys == ORC_LINSYS_DIRECT && w->ls.Lp) ? w->ls.Lp[w->ls.N] : 0)
#undef RET
  return NAN;
}
void orc_lp_set_trace(OrcWork *w, abip_int T, abip_float *buf) { w->trace_T = T; w->trace_buf = buf; w->trace_n = 0; }
abip_int orc_lp_trace_count(const OrcWork *w) { return w->trace_n; }
