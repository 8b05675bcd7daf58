/*
 * abip_hip_mex.c -- Matlab gateway for libabip_hip.so (LP path).
 *
 *     [x, y, s, info] = abip_direct(data, params)      built without -DABIP_HIP_PCG
 *     [x, y, s, info] = abip_indirect(data, params)    built with    -DABIP_HIP_PCG
 *
 * Same calling convention, recognised `params` fields and `info` fields as the reference's gateway
 * (src/abip-lp/mexfile/abip_mex.c:83-424), so scripts/matlab/abip_lpsolve.m calls it unchanged.  The reference's own gateway
 * also works against the library; this file exists so that the repository carries a gateway that
 * needs no reference source at all.  It cannot be built here (no Matlab); tests/test_mex_cpu.py compile-checks it against a
 * mock mex.h.
 *
 *   mex -largeArrayDims -I<repo>/include mex/abip_hip_mex.c -L<repo>/abip_amd/lib -labip_hip -output abip_direct
 */
#include <string.h>

#include "mex.h"
#include "abip.h"
#include "abip_hip.h"

static double get_field_or(const mxArray *s, const char *name, double dflt, int *present) {
  const mxArray *f = mxGetField(s, 0, name);
  if (present) *present = (f != NULL);
  return f ? *mxGetPr(f) : dflt;
}

static int parse_warm_start(const mxArray *p_mex, abip_float **p, abip_int len) { /* abip_mex.c:36-65 */
  *p = (abip_float *)mxCalloc(len, sizeof(abip_float));
  if (p_mex == NULL) return 0;
  if (mxIsSparse(p_mex)) { mexPrintf("Error in warm-start (the input vectors should be dense); the initial point is zero.\n"); return 0; }
  if (mxGetNumberOfElements(p_mex) != (size_t)len) { mexPrintf("Error in warm-start (the input vectors are of wrong size); the initial point is zero.\n"); return 0; }
  memcpy(*p, mxGetPr(p_mex), len * sizeof(abip_float));
  return 1;
}

static void set_output(mxArray **out, abip_float *v, abip_int len) { /* abip_mex.c:67-78 */
  *out = mxCreateDoubleMatrix(0, 0, mxREAL);
  mxSetPr(*out, v);
  mxSetM(*out, len);
  mxSetN(*out, 1);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  static const char *info_fields[] = {"status", "ipm_iter", "admm_iter", "mu", "pobj", "dobj", "resPri", "resDual", "relGap", "resInfeas", "resUnbdd", "setupTime", "solveTime"};
  if (nrhs != 2) mexErrMsgTxt("Inputs are required in this order: data struct, settings struct");
  if (nlhs > 4) mexErrMsgTxt("ABIP returns up to 4 output arguments only.");
  const mxArray *data = prhs[0], *settings = prhs[1];
  const mxArray *A_mex = mxGetField(data, 0, "A"), *b_mex = mxGetField(data, 0, "b"), *c_mex = mxGetField(data, 0, "c");
  if (A_mex == NULL) mexErrMsgTxt("ABIPData struct must contain a matrix 'A'.");
  if (!mxIsSparse(A_mex)) mexErrMsgTxt("Input matrix A must be in sparse format.");
  if (b_mex == NULL) mexErrMsgTxt("ABIPData struct must contain a vector 'b'.");
  if (mxIsSparse(b_mex)) mexErrMsgTxt("Input vector b must be in dense format.");
  if (c_mex == NULL) mexErrMsgTxt("ABIPData struct must contain a vector 'c'.");
  if (mxIsSparse(c_mex)) mexErrMsgTxt("Input vector c must be in dense format.");

  ABIPData d;
  ABIPSettings stgs;
  ABIPMatrix A;
  ABIPSolution sol = {0, 0, 0};
  ABIPInfo info;
  d.stgs = &stgs;
  d.n = (abip_int)mxGetNumberOfElements(c_mex);
  d.m = (abip_int)mxGetNumberOfElements(b_mex);
  d.b = (abip_float *)mxGetPr(b_mex);
  d.c = (abip_float *)mxGetPr(c_mex);
  abip_set_default_settings(&d);

  /* field names exactly as the reference reads them (abip_mex.c:183-341) */
  int has;
  double v;
#define INT_FIELD(name, member) v = get_field_or(settings, name, 0, &has); if (has) stgs.member = (abip_int)v
#define FLT_FIELD(name, member) v = get_field_or(settings, name, 0, &has); if (has) stgs.member = (abip_float)v
  INT_FIELD("max_ipm_iters", max_ipm_iters); INT_FIELD("max_admm_iters", max_admm_iters); FLT_FIELD("eps", eps);
  FLT_FIELD("cg_rate", cg_rate); FLT_FIELD("alpha", alpha); FLT_FIELD("rho_y", rho_y); INT_FIELD("normalize", normalize);
  FLT_FIELD("scale", scale); FLT_FIELD("sparsity_ratio", sparsity_ratio); INT_FIELD("adaptive", adaptive);
  INT_FIELD("adaptive_lookback", adaptive_lookback); FLT_FIELD("dynamic_sigma", dynamic_sigma); FLT_FIELD("dynamic_x", dynamic_x);
  FLT_FIELD("dynamic_eta", dynamic_eta); INT_FIELD("restart_thresh", restart_thresh); INT_FIELD("restart_fre", restart_fre);
  INT_FIELD("origin_rescale", origin_rescale); INT_FIELD("pc_ruiz_rescale", pc_ruiz_rescale); INT_FIELD("qp_rescale", qp_rescale);
  INT_FIELD("ruiz_iter", ruiz_iter); INT_FIELD("hybrid_mu", hybrid_mu); INT_FIELD("half_update", half_update);
  INT_FIELD("avg_criterion", avg_criterion); FLT_FIELD("hybrid_thresh", hybrid_thresh); FLT_FIELD("dynamic_sigma_second", dynamic_sigma_second);
  INT_FIELD("verbose", verbose);
  stgs.max_time = get_field_or(settings, "timelimit", 3600, NULL);   /* abip_mex.c:320-326 */
  stgs.pfeasopt = (abip_int)get_field_or(settings, "feasopt", 0, NULL); /* :334-341 */

  /* the library scales A in place (no COPYAMATRIX): hand it a private copy of Matlab's arrays */
  A.m = d.m; A.n = d.n;
  const mwIndex *jc = mxGetJc(A_mex), *ir = mxGetIr(A_mex);
  const abip_int nnz = (abip_int)jc[A.n];
  A.p = (abip_int *)mxMalloc(sizeof(abip_int) * (A.n + 1));
  A.i = (abip_int *)mxMalloc(sizeof(abip_int) * (nnz > 0 ? nnz : 1));
  A.x = (abip_float *)mxMalloc(sizeof(abip_float) * (nnz > 0 ? nnz : 1));
  for (abip_int k = 0; k <= A.n; ++k) A.p[k] = (abip_int)jc[k];
  for (abip_int k = 0; k < nnz; ++k) { A.i[k] = (abip_int)ir[k]; A.x[k] = mxGetPr(A_mex)[k]; }
  d.A = &A;
  d.sp = (abip_float)nnz / ((abip_float)A.m * (abip_float)A.n); /* abip_mex.c:362 */

  stgs.warm_start = parse_warm_start(mxGetField(data, 0, "x"), &sol.x, d.n);
  stgs.warm_start |= parse_warm_start(mxGetField(data, 0, "y"), &sol.y, d.m);
  stgs.warm_start |= parse_warm_start(mxGetField(data, 0, "s"), &sol.s, d.n);

#ifdef ABIP_HIP_PCG
  abip_hip_set_linsys(ABIP_HIP_LINSYS_INDIRECT);
#else
  abip_hip_set_linsys(ABIP_HIP_LINSYS_DIRECT);
#endif
  abip_main(&d, &sol, &info);

  set_output(&plhs[0], sol.x, d.n);
  if (nlhs > 1) set_output(&plhs[1], sol.y, d.m);
  if (nlhs > 2) set_output(&plhs[2], sol.s, d.n);
  if (nlhs > 3) {
    plhs[3] = mxCreateStructMatrix(1, 1, 13, info_fields);
    mxSetField(plhs[3], 0, "status", mxCreateString(info.status));
    mxSetField(plhs[3], 0, "ipm_iter", mxCreateDoubleScalar((double)info.ipm_iter));
    mxSetField(plhs[3], 0, "admm_iter", mxCreateDoubleScalar((double)info.admm_iter));
    mxSetField(plhs[3], 0, "pobj", mxCreateDoubleScalar(info.pobj));
    mxSetField(plhs[3], 0, "dobj", mxCreateDoubleScalar(info.dobj));
    mxSetField(plhs[3], 0, "resPri", mxCreateDoubleScalar(info.res_pri));
    mxSetField(plhs[3], 0, "resDual", mxCreateDoubleScalar(info.res_dual));
    mxSetField(plhs[3], 0, "relGap", mxCreateDoubleScalar(info.rel_gap));
    mxSetField(plhs[3], 0, "resInfeas", mxCreateDoubleScalar(info.res_infeas));
    mxSetField(plhs[3], 0, "resUnbdd", mxCreateDoubleScalar(info.res_unbdd));
    mxSetField(plhs[3], 0, "setupTime", mxCreateDoubleScalar(info.setup_time));
    mxSetField(plhs[3], 0, "solveTime", mxCreateDoubleScalar(info.solve_time));
  }
  mxFree(A.p); mxFree(A.i); mxFree(A.x);
}
