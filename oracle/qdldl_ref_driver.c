/* oracle/qdldl_ref_driver.c -- TEST INFRASTRUCTURE.  A caller of the reference's vendored QDLDL (src/external/qdldl/src/qdldl.c, compiled from where it
 * lies by `make ref` together with this file into oracle/_ref/libqdldl_ref.so): factor a quasi-definite K given by its upper triangle (CSC, diagonal
 * present, natural order) and solve K x = b, with the calls and work arrays of the reference's own use (src/abip-qcp/source/linsys.c:560-625, 310-316).
 * tests/golden/make_golden_qdldl.py turns its output into the committed fixtures tests/golden/qdldl_*.npz. */
#include <stdlib.h>
#include "qdldl.h"

int qdldl_ref_solve(int n, const int *Ap, const int *Ai, const double *Ax, double *b /* in: rhs, out: solution */, double *Dout /* n pivots, may be NULL */) {
  QDLDL_int *etree = (QDLDL_int *)malloc(sizeof(QDLDL_int) * n), *Lnz = (QDLDL_int *)malloc(sizeof(QDLDL_int) * n);
  QDLDL_int *iwork = (QDLDL_int *)malloc(sizeof(QDLDL_int) * 3 * n), *Lp = (QDLDL_int *)malloc(sizeof(QDLDL_int) * (n + 1)), *Li = NULL;
  QDLDL_bool *bwork = (QDLDL_bool *)malloc(sizeof(QDLDL_bool) * n);
  QDLDL_float *fwork = (QDLDL_float *)malloc(sizeof(QDLDL_float) * n), *D = (QDLDL_float *)malloc(sizeof(QDLDL_float) * n), *Dinv = (QDLDL_float *)malloc(sizeof(QDLDL_float) * n), *Lx = NULL;
  int rc = -1;
  const QDLDL_int sumLnz = QDLDL_etree(n, Ap, Ai, iwork, Lnz, etree);
  if (sumLnz >= 0) {
    Li = (QDLDL_int *)malloc(sizeof(QDLDL_int) * (sumLnz > 0 ? sumLnz : 1));
    Lx = (QDLDL_float *)malloc(sizeof(QDLDL_float) * (sumLnz > 0 ? sumLnz : 1));
    if (QDLDL_factor(n, Ap, Ai, Ax, Lp, Li, Lx, D, Dinv, Lnz, etree, bwork, iwork, fwork) >= 0) {
      QDLDL_solve(n, Lp, Li, Lx, Dinv, b);
      if (Dout) for (int i = 0; i < n; ++i) Dout[i] = D[i];
      rc = 0;
    }
  }
  free(etree); free(Lnz); free(iwork); free(Lp); free(Li); free(bwork); free(fwork); free(D); free(Dinv); free(Lx);
  return rc;
}

/* The kernel-level CPU leg of the conic direct back-end: factor once, then time `nrhs` calls of the reference's QDLDL_solve (qdldl.c:236-281,
 * one thread) on right-hand sides B (nrhs x n, row-major; overwritten by the solutions).  out[0] = seconds in QDLDL_factor, out[1] = seconds in the nrhs solves,
 * out[2] = non-zeros of L.  scripts/qdldl_cpu_leg.py calls it on the C5 KKT system. */
#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int qdldl_ref_time_solves(int n, const int *Ap, const int *Ai, const double *Ax, int nrhs, double *B, double *out) {
  QDLDL_int *etree = (QDLDL_int *)malloc(sizeof(QDLDL_int) * n), *Lnz = (QDLDL_int *)malloc(sizeof(QDLDL_int) * n);
  QDLDL_int *iwork = (QDLDL_int *)malloc(sizeof(QDLDL_int) * 3 * n), *Lp = (QDLDL_int *)malloc(sizeof(QDLDL_int) * (n + 1)), *Li = NULL;
  QDLDL_bool *bwork = (QDLDL_bool *)malloc(sizeof(QDLDL_bool) * n);
  QDLDL_float *fwork = (QDLDL_float *)malloc(sizeof(QDLDL_float) * n), *D = (QDLDL_float *)malloc(sizeof(QDLDL_float) * n), *Dinv = (QDLDL_float *)malloc(sizeof(QDLDL_float) * n), *Lx = NULL;
  int rc = -1;
  const QDLDL_int sumLnz = QDLDL_etree(n, Ap, Ai, iwork, Lnz, etree);
  if (sumLnz >= 0) {
    Li = (QDLDL_int *)malloc(sizeof(QDLDL_int) * (size_t)(sumLnz > 0 ? sumLnz : 1));
    Lx = (QDLDL_float *)malloc(sizeof(QDLDL_float) * (size_t)(sumLnz > 0 ? sumLnz : 1));
    double t0 = now_s();
    if (Li && Lx && QDLDL_factor(n, Ap, Ai, Ax, Lp, Li, Lx, D, Dinv, Lnz, etree, bwork, iwork, fwork) >= 0) {
      out[0] = now_s() - t0;
      t0 = now_s();
      for (int k = 0; k < nrhs; ++k) QDLDL_solve(n, Lp, Li, Lx, Dinv, B + (size_t)k * n);
      out[1] = now_s() - t0;
      out[2] = (double)sumLnz;
      rc = 0;
    }
  }
  free(etree); free(Lnz); free(iwork); free(Lp); free(Li); free(bwork); free(fwork); free(D); free(Dinv); free(Lx);
  return rc;
}
