/* TEST PROGRAM (tests/test_gpu_cprog.py): a plain C caller of the reference's entry points as INTEGRATION.md section 2 shows them --
 * abip_main, and abip_init / abip_solve x2 / abip_finish with new (b, c) between the two solves (the reference's contract,
 * src/abip-lp/include/abip.h:116-121: "solve: can be called many times with different b,c per init call").
 *
 *   abip_c_prog <problem.bin> <out.bin> <mode: main|resolve> <linsys: 0 direct | 1 indirect> <eps>
 * problem.bin: long m, n, nnz; long Ap[n+1]; long Ai[nnz]; double Ax[nnz], b[m], c[n], b2[m], c2[n]
 * out.bin    : per solve: double info[8] = {status_val, ipm_iter, admm_iter, pobj, dobj, res_pri, res_dual, rel_gap}, x[n], y[m], s[n] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "abip.h"
#include "abip_hip.h"

static void *rd(FILE *f, size_t bytes) { void *p = malloc(bytes ? bytes : 1); if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(2); } return p; }
static void dump(FILE *o, const ABIPInfo *info, const ABIPSolution *sol, abip_int m, abip_int n) {
  const double v[8] = {(double)info->status_val, (double)info->ipm_iter, (double)info->admm_iter, info->pobj, info->dobj, info->res_pri, info->res_dual, info->rel_gap};
  fwrite(v, sizeof(double), 8, o);
  fwrite(sol->x, sizeof(double), n, o); fwrite(sol->y, sizeof(double), m, o); fwrite(sol->s, sizeof(double), n, o);
}

int main(int argc, char **argv) {
  if (argc != 6) { fprintf(stderr, "usage\n"); return 2; }
  FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
  long hdr[3]; if (fread(hdr, sizeof(long), 3, f) != 3) return 2;
  const abip_int m = hdr[0], n = hdr[1], nnz = hdr[2];
  ABIPMatrix A; A.m = m; A.n = n;
  A.p = (abip_int *)rd(f, sizeof(long) * (n + 1)); A.i = (abip_int *)rd(f, sizeof(long) * nnz); A.x = (abip_float *)rd(f, sizeof(double) * nnz);
  double *b = (double *)rd(f, 8 * m), *c = (double *)rd(f, 8 * n), *b2 = (double *)rd(f, 8 * m), *c2 = (double *)rd(f, 8 * n);
  fclose(f);
  ABIPSettings stgs;
  ABIPData d = {m, n, &A, b, c, (double)nnz / ((double)m * (double)n), &stgs};
  abip_set_default_settings(&d);
  stgs.max_time = 3600; stgs.pfeasopt = 0; /* the mex sets these two, abip_mex.c:320-341 */
  stgs.eps = atof(argv[5]); stgs.verbose = 0;
  abip_hip_set_linsys(atoi(argv[4]) ? ABIP_HIP_LINSYS_INDIRECT : ABIP_HIP_LINSYS_DIRECT);
  FILE *o = fopen(argv[2], "wb"); if (!o) return 2;
  ABIPSolution sol = {0, 0, 0};
  ABIPInfo info;
  memset(&info, 0, sizeof(info));
  if (!strcmp(argv[3], "main")) {
    const abip_int st = abip_main(&d, &sol, &info); /* sol.x / y / s are malloc'ed by the library, freed by the caller */
    if (st != info.status_val) return 3;
    dump(o, &info, &sol, m, n);
  } else {
    ABIPWork *w = abip_init(&d, &info);
    if (!w) return 4;
    abip_solve(w, &d, &sol, &info);
    dump(o, &info, &sol, m, n);
    d.b = b2; d.c = c2;                    /* new right-hand side and cost, same A, same work */
    /* the solver WRITES three settings while it runs, here exactly as in the reference (abip.c:2042/2048 avg_criterion, :2254 dynamic_sigma,
     * :2220 max_admm_iters; w->stgs aliases d->stgs, abip.c:1760).  Left as the first solve leaves them, dynamic_sigma = dynamic_sigma_second
     * > 0 disables the first-phase mu rule (abip.c:2251-2277) and the second solve never leaves mu = 1 -- upstream too.  A caller that
     * re-solves puts them back: */
    stgs.dynamic_sigma = -1.0; stgs.avg_criterion = 0; stgs.max_admm_iters = 1000000;
    abip_solve(w, &d, &sol, &info);
    dump(o, &info, &sol, m, n);
    abip_finish(w);
  }
  fclose(o);
  free(sol.x); free(sol.y); free(sol.s);
  printf("%s\n", abip_version());
  return 0;
}
