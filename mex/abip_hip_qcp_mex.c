/*
 * abip_hip_qcp_mex.c -- Matlab gateways for the conic path of libabip_hip.so.
 *
 *     [sol, info] = abip_qcp(data, cones, settings)     built without -DABIP_HIP_ML     (src/abip-qcp/mex/abip_qcp_mex.c:109-525)
 *     [sol, info] = abip_ml(data, settings)             built with    -DABIP_HIP_ML     (src/abip-qcp/mex/abip_ml_mex.c:90-449)
 *
 * Same inputs, recognised `settings` fields, output structs and `info` fields as the reference's gateways, so scripts/matlab/abip_qcpsolve.m
 * and scripts/bench-qcp/test_{lasso,svm}.m call them unchanged.  The reference's two gateways cannot be compiled against this library as they
 * stand: they include its linsys.h / cones.h, which include MKL headers.
 *   abip_qcp: data.{A (sparse), Q (sparse, optional), b, c}; cones.{q, rq, f, z, l}; sol.{x, y, s}.
 *   abip_ml:  data.{X (sparse), y, lambda}; settings.prob_type mandatory: 0 LASSO (sol.x = beta), 1 SVM-SOCP, 3 SVM-QP (sol.{w, b, xi}; the
 *             reference's output switch compares prob_type with 2 and 4, abip_ml_mex.c:362, and therefore hands Matlab {x: w} for the SVMs --
 *             here x is returned as well, so scripts written against either read what they expect).
 *
 *   mex -largeArrayDims -I<repo>/include mex/abip_hip_qcp_mex.c -L<repo>/abip_amd/lib -labip_hip -output abip_qcp
 */
#include <stdlib.h>
#include <string.h>

#include "mex.h"
#include "abip_qcp.h"

static qcp_int *to_int(const mwIndex *src, size_t n) { /* cast_to_abip_int_arr, abip_qcp_mex.c:52-66 */
  qcp_int *out = (qcp_int *)mxMalloc(sizeof(qcp_int) * (n ? n : 1));
  for (size_t k = 0; k < n; ++k) out[k] = (qcp_int)src[k];
  return out;
}
static size_t max_dim(const mxArray *a) { const size_t m = mxGetM(a), n = mxGetN(a); return m > n ? m : n; }
static void load_sparse(const mxArray *M, QCPMatrix *out) {
  out->m = (qcp_int)mxGetM(M); out->n = (qcp_int)mxGetN(M);
  out->p = to_int(mxGetJc(M), (size_t)out->n + 1);
  out->i = to_int(mxGetIr(M), (size_t)out->p[out->n]);
  out->x = mxGetPr(M); /* read-only inside the library (it scales private copies) */
}
static void set_field_vec(mxArray *s, const char *name, const qcp_float *v, size_t len) {
  mxArray *a = mxCreateDoubleMatrix(len, 1, mxREAL);
  if (len) memcpy(mxGetPr(a), v, sizeof(double) * len);
  mxSetField(s, 0, name, a);
}
static void read_settings(const mxArray *settings, QCPSettings *st) { /* abip_qcp_mex.c:296-434 = abip_ml_mex.c:160-312 */
  const mxArray *t;
#define FLT(name) if ((t = mxGetField(settings, 0, #name)) != NULL) st->name = (qcp_float)*mxGetPr(t)
#define INT(name) if ((t = mxGetField(settings, 0, #name)) != NULL) st->name = (qcp_int)*mxGetPr(t)
  FLT(alpha); FLT(cg_rate);
  if ((t = mxGetField(settings, 0, "eps")) != NULL) { st->eps = (qcp_float)*mxGetPr(t); st->eps_p = st->eps_d = st->eps_g = st->eps_inf = st->eps_unb = st->eps; }
  FLT(eps_p); FLT(eps_d); FLT(eps_g); FLT(eps_inf); FLT(eps_unb);
  INT(max_admm_iters); INT(max_ipm_iters); INT(normalize); FLT(rho_y); FLT(rho_x); FLT(rho_tau);
  if ((t = mxGetField(settings, 0, "scale")) != NULL) st->scale = (qcp_float)(qcp_int)*mxGetPr(t); /* sic: read through an integer cast (:369) */
  INT(scale_bc); INT(scale_E); INT(use_indirect); INT(verbose); INT(linsys_solver); INT(inner_check_period); INT(outer_check_period);
  FLT(err_dif); FLT(time_limit); FLT(psi); INT(origin_scaling); INT(ruiz_scaling); INT(pc_scaling);
#undef FLT
#undef INT
}
static mxArray *make_info(const QCPInfo *info) { /* abip_qcp_mex.c:455-513 (times in seconds) */
  static const char *info_fields[] = {"ipm_iter", "admm_iter", "status", "pobj", "dobj", "res_pri", "res_dual", "gap", "status_val", "setup_time", "solve_time",
                                      "runtime", "lin_sys_time_per_iter", "avg_cg_iters"};
  mxArray *o = mxCreateStructMatrix(1, 1, 14, info_fields);
  mxSetField(o, 0, "status", mxCreateString(info->status));
  mxSetField(o, 0, "ipm_iter", mxCreateDoubleScalar((double)info->ipm_iter));
  mxSetField(o, 0, "admm_iter", mxCreateDoubleScalar((double)info->admm_iter));
  mxSetField(o, 0, "status_val", mxCreateDoubleScalar((double)info->status_val));
  mxSetField(o, 0, "pobj", mxCreateDoubleScalar(info->pobj));
  mxSetField(o, 0, "dobj", mxCreateDoubleScalar(info->dobj));
  mxSetField(o, 0, "res_pri", mxCreateDoubleScalar(info->res_pri));
  mxSetField(o, 0, "res_dual", mxCreateDoubleScalar(info->res_dual));
  mxSetField(o, 0, "gap", mxCreateDoubleScalar(info->rel_gap));
  mxSetField(o, 0, "setup_time", mxCreateDoubleScalar(info->setup_time / 1e3));
  mxSetField(o, 0, "solve_time", mxCreateDoubleScalar(info->solve_time / 1e3));
  mxSetField(o, 0, "runtime", mxCreateDoubleScalar((info->solve_time + info->setup_time) / 1e3));
  mxSetField(o, 0, "lin_sys_time_per_iter", mxCreateDoubleScalar(info->avg_linsys_time / 1e3));
  mxSetField(o, 0, "avg_cg_iters", mxCreateDoubleScalar(info->avg_cg_iters));
  return o;
}
#ifndef ABIP_HIP_ML
static qcp_int *cone_list(const mxArray *f, qcp_int *count) { /* abip_qcp_mex.c:230-262 */
  *count = 0;
  if (!f || mxIsEmpty(f)) return NULL;
  const mwSize *dims = mxGetDimensions(f);
  qcp_int len = (qcp_int)dims[0];
  if (mxGetNumberOfDimensions(f) > 1 && dims[0] == 1) len = (qcp_int)dims[1];
  qcp_int *out = (qcp_int *)mxMalloc(sizeof(qcp_int) * (len > 0 ? len : 1));
  for (qcp_int i = 0; i < len; ++i) out[i] = (qcp_int)mxGetPr(f)[i];
  *count = len;
  return out;
}
static qcp_int cone_scalar(const mxArray *f) { return (f && !mxIsEmpty(f)) ? (qcp_int)*mxGetPr(f) : 0; }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  static const char *sol_fields[] = {"x", "y", "s"};
  if (nrhs != 3) mexErrMsgTxt("Inputs are required in this order: data struct, cone struct, settings struct");
  if (nlhs > 2) mexErrMsgTxt("abip_qcp returns up to 2 output arguments only.");
  const mxArray *data = prhs[0], *cone = prhs[1], *settings = prhs[2];
  const mxArray *A_mex = mxGetField(data, 0, "A"), *Q_mex = mxGetField(data, 0, "Q"), *b_mex = mxGetField(data, 0, "b"), *c_mex = mxGetField(data, 0, "c");
  if (c_mex == NULL) mexErrMsgTxt("ABIPData struct must contain a `c` entry.");
  if (mxIsSparse(c_mex)) mexErrMsgTxt("Input vector c must be in dense format (pass in full(c))");
  if (b_mex != NULL && mxIsSparse(b_mex)) mexErrMsgTxt("Input vector b must be in dense format (pass in full(b))");
  if (A_mex != NULL && !mxIsSparse(A_mex)) mexErrMsgTxt("Input matrix A must be in sparse format (pass in sparse(A))");
  if (Q_mex != NULL && !mxIsSparse(Q_mex)) mexErrMsgTxt("Input matrix Q must be in sparse format (pass in sparse(Q))");
  if (A_mex == NULL || b_mex == NULL) mexErrMsgTxt("the device path needs `A` and `b`"); /* (the reference goes on with m = 0, abip_qcp_mex.c:176-183) */
  QCPData d; QCPSettings stgs; QCPMatrix A, Q; QCPCone K; QCPSolution sol = {0, 0, 0}; QCPInfo info;
  memset(&d, 0, sizeof(d)); memset(&info, 0, sizeof(info));
  d.stgs = &stgs;
  load_sparse(A_mex, &A); d.A = &A; d.m = A.m; d.n = A.n;
  if (Q_mex) { load_sparse(Q_mex, &Q); d.Q = &Q; }
  if (max_dim(b_mex) != (size_t)d.m || max_dim(c_mex) != (size_t)d.n) mexErrMsgTxt("b and c must have as many entries as A has rows and columns");
  d.b = mxGetPr(b_mex); d.c = mxGetPr(c_mex);
  K.q = cone_list(mxGetField(cone, 0, "q"), &K.qsize); K.rq = cone_list(mxGetField(cone, 0, "rq"), &K.rqsize);
  K.f = cone_scalar(mxGetField(cone, 0, "f")); K.z = cone_scalar(mxGetField(cone, 0, "z")); K.l = cone_scalar(mxGetField(cone, 0, "l"));
  abip_qcp_set_default_settings(&d);
  read_settings(settings, &stgs);
  stgs.prob_type = 2; /* enum QCP, abip_qcp_mex.c:436 */
  abip_qcp(&d, &sol, &info, &K);
  plhs[0] = mxCreateStructMatrix(1, 1, 3, sol_fields);
  const int ok = sol.x && sol.y && sol.s;
  set_field_vec(plhs[0], "x", sol.x, ok ? (size_t)d.n : 0); set_field_vec(plhs[0], "y", sol.y, ok ? (size_t)d.m : 0); set_field_vec(plhs[0], "s", sol.s, ok ? (size_t)d.n : 0);
  if (nlhs > 1) plhs[1] = make_info(&info);
  free(sol.x); free(sol.y); free(sol.s); /* malloc'ed by the library (abip.c:452-476) */
  mxFree(A.p); mxFree(A.i); if (Q_mex) { mxFree(Q.p); mxFree(Q.i); } if (K.q) mxFree(K.q); if (K.rq) mxFree(K.rq);
}
#else
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  static const char *svm_fields[] = {"w", "b", "xi", "x"};
  static const char *lasso_fields[] = {"x"};
  if (nrhs != 2) mexErrMsgTxt("Inputs are required in this order: data struct, settings struct");
  if (nlhs > 2) mexErrMsgTxt("abip_ml returns up to 2 output arguments only.");
  const mxArray *data = prhs[0], *settings = prhs[1];
  const mxArray *X_mex = mxGetField(data, 0, "X"), *y_mex = mxGetField(data, 0, "y"), *l_mex = mxGetField(data, 0, "lambda");
  if (X_mex == NULL) mexErrMsgTxt("ABIPData struct must contain a `X` entry.");
  if (!mxIsSparse(X_mex)) mexErrMsgTxt("Input matrix X must be in sparse format (pass in sparse(X))");
  if (y_mex == NULL) mexErrMsgTxt("ABIPData struct must contain a `y` entry.");
  if (mxIsSparse(y_mex)) mexErrMsgTxt("Input vector y must be in dense format (pass in full(y))");
  if (l_mex == NULL) mexErrMsgTxt("ABIPData struct must contain a `lambda` entry.");
  QCPData d; QCPSettings stgs; QCPMatrix X; QCPCone K; QCPSolution sol = {0, 0, 0}; QCPInfo info;
  memset(&d, 0, sizeof(d)); memset(&info, 0, sizeof(info));
  d.stgs = &stgs;
  load_sparse(X_mex, &X); d.A = &X; d.m = X.m; d.n = X.n;
  if (max_dim(y_mex) != (size_t)d.m) mexErrMsgTxt("y must have as many entries as X has rows");
  d.b = mxGetPr(y_mex); d.lambda = (qcp_float)*mxGetPr(l_mex);
  abip_qcp_set_default_settings(&d);
  read_settings(settings, &stgs);
  const mxArray *pt = mxGetField(settings, 0, "prob_type"); /* abip_ml_mex.c:266-276 */
  if (pt == NULL) mexErrMsgTxt("Please input the machine learning problem type");
  stgs.prob_type = (qcp_int)*mxGetPr(pt);
  if (stgs.prob_type != 0 && stgs.prob_type != 1 && stgs.prob_type != 3) mexErrMsgTxt("Invalid problem type");
  qcp_int rq = 0;
  K.q = NULL; K.qsize = 0; K.rq = &rq; K.rqsize = 1; K.f = 0; K.z = 0; K.l = 0; /* :315-342 */
  if (stgs.prob_type == 0) { rq = 2 + d.m; K.l = 2 * d.n; }
  else if (stgs.prob_type == 1) { rq = 2 + d.n; K.l = 2 + 2 * d.m + 2 * d.n; }
  else { K.rq = NULL; K.rqsize = 0; K.f = d.n + 1; K.l = 2 * d.m; }
  abip_qcp(&d, &sol, &info, &K);
  if (stgs.prob_type == 0) {
    plhs[0] = mxCreateStructMatrix(1, 1, 1, lasso_fields);
    set_field_vec(plhs[0], "x", sol.x, sol.x ? (size_t)d.n : 0);
  } else {
    const int ok = sol.x && sol.y && sol.s;
    plhs[0] = mxCreateStructMatrix(1, 1, 4, svm_fields);
    set_field_vec(plhs[0], "w", sol.x, ok ? (size_t)d.n : 0); set_field_vec(plhs[0], "b", sol.y, ok ? 1 : 0); set_field_vec(plhs[0], "xi", sol.s, ok ? (size_t)d.m : 0);
    set_field_vec(plhs[0], "x", sol.x, ok ? (size_t)d.n : 0);
  }
  if (nlhs > 1) plhs[1] = make_info(&info);
  free(sol.x); free(sol.y); free(sol.s);
  mxFree(X.p); mxFree(X.i);
}
#endif
