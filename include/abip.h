/*
 * abip.h -- public C ABI of libabip_hip.so, the MI355X-native ABIP-LP solver.
 *
 * This header is the drop-in boundary: every struct below is layout-compatible
 * with the reference's (leavesgrp/ABIP v2.0.0) public structs and every entry
 * point has the reference's name, argument meaning and error behaviour, so a
 * program (or the Matlab mex gateway src/abip-lp/mexfile/abip_mex.c:83-424)
 * written against the reference's src/abip-lp/include/abip.h links against this
 * library unchanged.  Nothing here mentions torch, HIP or device pointers: the
 * caller hands over host memory exactly as it does to the reference.
 *
 * Reference interface each item replaces:
 *   abip_int / abip_float ........ src/abip-lp/include/glbopts.h:86-112
 *   status codes ................. src/abip-lp/include/glbopts.h:22-31
 *   ABIPMatrix (CSC) ............. src/abip-lp/linsys/amatrix.h:10-17
 *   ABIPData ..................... src/abip-lp/include/abip.h:23-34
 *   ABIPSettings ................. src/abip-lp/include/abip.h:36-79
 *   ABIPSolution ................. src/abip-lp/include/abip.h:81-86
 *   ABIPInfo ..................... src/abip-lp/include/abip.h:88-105
 *   abip_init/solve/finish/main .. src/abip-lp/include/abip.h:119-124
 *   abip_version ................. src/abip-lp/include/abip.h:124
 *   abip_set_default_settings,
 *   abip_free_data, abip_free_sol  src/abip-lp/include/util.h:58-60
 *
 * The reference selects the KKT solver at LINK time (direct.c vs indirect.c are
 * two builds of the same symbols; the Matlab front end picks abip_direct or
 * abip_indirect from params.pcg, scripts/matlab/abip_lpsolve.m:16-21).  This
 * library carries both device back-ends; abip_hip_set_linsys() below (or the
 * environment variable ABIP_HIP_LINSYS=direct|indirect) picks one, default
 * direct exactly like params.pcg = 0.
 */
#ifndef ABIP_HIP_ABIP_H
#define ABIP_HIP_ABIP_H

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these declarations ARE its export list */
#endif

/* The mex build of the reference defines DLONG (make_abip.m:50-54); so do we
 * unless the integrator asks for 32-bit indices with -DABIP_INT32. */
#ifndef ABIP_INT32
typedef long abip_int;
#else
typedef int abip_int;
#endif
typedef double abip_float;

#define ABIP_VERSION "2.0.0-mi355x"

#define ABIP_INFEASIBLE_INACCURATE (-7)
#define ABIP_UNBOUNDED_INACCURATE (-6)
#define ABIP_SIGINT (-5)
#define ABIP_FAILED (-4)
#define ABIP_INDETERMINATE (-3)
#define ABIP_INFEASIBLE (-2)
#define ABIP_UNBOUNDED (-1)
#define ABIP_UNFINISHED (0)
#define ABIP_SOLVED (1)
#define ABIP_SOLVED_INACCURATE (2)

typedef struct ABIP_A_DATA_MATRIX ABIPMatrix;
typedef struct ABIP_PROBLEM_DATA ABIPData;
typedef struct ABIP_SETTINGS ABIPSettings;
typedef struct ABIP_SOL_VARS ABIPSolution;
typedef struct ABIP_INFO ABIPInfo;
typedef struct ABIP_WORK ABIPWork; /* opaque to callers */

/* A in compressed-sparse-column form, row indices ascending within a column. */
struct ABIP_A_DATA_MATRIX {
  abip_float *x; /* values,       size p[n]  */
  abip_int *i;   /* row indices,  size p[n]  */
  abip_int *p;   /* column starts, size n+1  */
  abip_int m;    /* rows    */
  abip_int n;    /* columns */
};

struct ABIP_PROBLEM_DATA {
  abip_int m;
  abip_int n;
  ABIPMatrix *A;
  abip_float *b;  /* size m */
  abip_float *c;  /* size n */
  abip_float sp;  /* nnz(A)/(m*n); the caller sets it (abip_mex.c:362) */
  ABIPSettings *stgs;
};

struct ABIP_SETTINGS {
  abip_int normalize;
  abip_int pfeasopt;
  abip_float scale;
  abip_float rho_y;
  abip_float sparsity_ratio;

  abip_int max_ipm_iters;
  abip_int max_admm_iters;
  abip_float max_time;

  abip_float eps;
  abip_float alpha;
  abip_float cg_rate;

  abip_int adaptive;
  abip_float eps_cor;
  abip_float eps_pen;

  abip_float dynamic_sigma;
  abip_float dynamic_x;
  abip_float dynamic_eta;

  abip_int restart_fre;
  abip_int restart_thresh;

  abip_int verbose;
  abip_int warm_start;

  abip_int adaptive_lookback;

  abip_int origin_rescale;
  abip_int pc_ruiz_rescale;
  abip_int qp_rescale;
  abip_int ruiz_iter;
  abip_int hybrid_mu;
  abip_float hybrid_thresh;
  abip_float dynamic_sigma_second;
  abip_int half_update;
  abip_int avg_criterion;
};

struct ABIP_SOL_VARS {
  abip_float *x; /* size n */
  abip_float *y; /* size m */
  abip_float *s; /* size n */
};

struct ABIP_INFO {
  char status[32];
  abip_int status_val;
  abip_int ipm_iter;
  abip_int admm_iter;

  abip_float pobj;
  abip_float dobj;
  abip_float res_pri;
  abip_float res_dual;
  abip_float rel_gap;
  abip_float res_infeas;
  abip_float res_unbdd;

  abip_float setup_time; /* ms */
  abip_float solve_time; /* ms */
};

/* ---- the reference's entry points (same names, same semantics) ------------
 * abip_init   : validate, scale A (in place, un-scaled again by abip_finish --
 *               the reference without COPYAMATRIX, abip.c:1799-1807,2310-2317),
 *               build the KKT back-end, upload everything to the GPU.
 *               NULL on validation / allocation / factorisation failure or when
 *               no usable HIP device exists (the library never falls back to a
 *               CPU path).
 * abip_solve  : run the ADMM-based interior-point iteration on the device;
 *               returns the status code, also in info->status_val.  sol->x/y/s
 *               are malloc'ed when NULL and owned by the caller afterwards.
 *               As in the reference, w->stgs aliases d->stgs and the solver
 *               writes stgs->avg_criterion, dynamic_sigma and max_admm_iters.
 * abip_finish : un-scale A, free host and device state.
 * abip_main   : init + solve + finish. */
ABIPWork *abip_init(const ABIPData *d, ABIPInfo *info);
abip_int abip_solve(ABIPWork *w, const ABIPData *d, ABIPSolution *sol, ABIPInfo *info);
void abip_finish(ABIPWork *w);
abip_int abip_main(const ABIPData *d, ABIPSolution *sol, ABIPInfo *info);
const char *abip_version(void);

void abip_set_default_settings(ABIPData *d); /* util.c:288-329; max_time/pfeasopt are the caller's, as in the mex */
void abip_free_data(ABIPData *d);
void abip_free_sol(ABIPSolution *sol);

/* ---- back-end selection (replaces the reference's two link-time variants) - */
#define ABIP_HIP_LINSYS_DIRECT 0   /* linsys/direct.c   : LDL' once on host, SpTRSV on device */
#define ABIP_HIP_LINSYS_INDIRECT 1 /* linsys/indirect.c : Jacobi-PCG on device               */
void abip_hip_set_linsys(int which);
int abip_hip_get_linsys(void);

/* ---- ownership of A (replaces the reference's COPYAMATRIX build switch, abip.c:1799-1807, make_abip.m:13) ----
 * on = 1 (default, also ABIP_HIP_COPYAMATRIX unset): abip_init scales a private copy of A's values; the caller's matrix is never written
 *          (what the reference's mex build does -- Matlab owns A).
 * on = 0 (or ABIP_HIP_COPYAMATRIX=0): A is scaled in place and un-scaled by abip_finish (abip.c:2310-2317), like the plain C build. */
void abip_hip_set_copy_a_matrix(int on); /* 1 copy (default), 0 scale the caller's A in place, < 0 back to ABIP_HIP_COPYAMATRIX / the default */
int abip_hip_get_copy_a_matrix(void);    /* what the program set: 1, 0, or -1 if it set nothing */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ABIP_HIP_ABIP_H */
