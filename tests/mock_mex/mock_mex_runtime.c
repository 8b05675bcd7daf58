/* TEST INFRASTRUCTURE -- a tiny stand-in for the Matlab runtime behind tests/mock_mex/mex.h, so that a mex gateway's mexFunction
 * can be EXECUTED by the test-suite (no Matlab in the image): dense / sparse double arrays, 1x1 structs, strings, mxMalloc,
 * and mexErrMsgTxt as a longjmp back into mock_call().  The test drives it through the mock_* helpers below (ctypes). */
#define _POSIX_C_SOURCE 200809L /* strdup */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

enum { K_DOUBLE = 0, K_SPARSE = 1, K_STRUCT = 2, K_STRING = 3 };
struct mxArray_tag {
  int kind;
  mwSize dims[2];
  double *pr;
  mwIndex *ir, *jc;
  int nfields;
  char **names;
  mxArray **vals;
  char *str;
};

static jmp_buf g_jmp;
static int g_jmp_armed = 0;
static char g_err[512];

static mxArray *new_arr(int kind, mwSize m, mwSize n) {
  mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
  a->kind = kind; a->dims[0] = m; a->dims[1] = n;
  return a;
}
void *mxMalloc(size_t n) { return malloc(n ? n : 1); }
void *mxCalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz ? sz : 1); }
void *mxRealloc(void *p, size_t n) { return realloc(p, n ? n : 1); }
void mxFree(void *p) { free(p); }
mxArray *mxGetField(const mxArray *s, mwIndex i, const char *name) {
  (void)i;
  if (!s || s->kind != K_STRUCT) return NULL;
  for (int f = 0; f < s->nfields; ++f) if (!strcmp(s->names[f], name)) return s->vals[f];
  return NULL;
}
double *mxGetPr(const mxArray *a) { return a ? a->pr : NULL; }
int mxIsSparse(const mxArray *a) { return a && a->kind == K_SPARSE; }
size_t mxGetNumberOfElements(const mxArray *a) { return a ? a->dims[0] * a->dims[1] : 0; }
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims; }
mwSize mxGetNumberOfDimensions(const mxArray *a) { (void)a; return 2; }
size_t mxGetM(const mxArray *a) { return a ? a->dims[0] : 0; }
size_t mxGetN(const mxArray *a) { return a ? a->dims[1] : 0; }
int mxIsEmpty(const mxArray *a) { return !a || a->dims[0] * a->dims[1] == 0; }
mwIndex *mxGetJc(const mxArray *a) { return a->jc; }
mwIndex *mxGetIr(const mxArray *a) { return a->ir; }
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) {
  (void)c;
  mxArray *a = new_arr(K_DOUBLE, m, n);
  a->pr = (double *)mxCalloc(m * n, sizeof(double));
  return a;
}
mxArray *mxCreateDoubleScalar(double v) { mxArray *a = mxCreateDoubleMatrix(1, 1, mxREAL); a->pr[0] = v; return a; }
mxArray *mxCreateString(const char *s) { mxArray *a = new_arr(K_STRING, 1, strlen(s)); a->str = strdup(s); return a; }
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names) {
  mxArray *a = new_arr(K_STRUCT, m, n);
  a->nfields = nfields;
  a->names = (char **)calloc(nfields > 0 ? nfields : 1, sizeof(char *));
  a->vals = (mxArray **)calloc(nfields > 0 ? nfields : 1, sizeof(mxArray *));
  for (int f = 0; f < nfields; ++f) a->names[f] = strdup(names[f]);
  return a;
}
mxArray *mxCreateStructArray(mwSize ndim, const mwSize *dims, int nfields, const char **names) {
  return mxCreateStructMatrix(ndim > 0 ? dims[0] : 1, ndim > 1 ? dims[1] : 1, nfields, names);
}
void mxSetField(mxArray *s, mwIndex i, const char *name, mxArray *v) {
  (void)i;
  for (int f = 0; f < s->nfields; ++f) if (!strcmp(s->names[f], name)) { s->vals[f] = v; return; }
}
void mxSetPr(mxArray *a, double *p) { if (a->pr && a->pr != p) free(a->pr); a->pr = p; }
void mxSetM(mxArray *a, mwSize m) { a->dims[0] = m; }
void mxSetN(mxArray *a, mwSize n) { a->dims[1] = n; }
int mexPrintf(const char *fmt, ...) { va_list ap; va_start(ap, fmt); const int r = vfprintf(stdout, fmt, ap); va_end(ap); return r; }
void mexErrMsgTxt(const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  if (g_jmp_armed) longjmp(g_jmp, 1);
  fprintf(stderr, "mexErrMsgTxt outside mock_call: %s\n", g_err);
  abort();
}

/* ---- what the test calls -------------------------------------------------------------------------------- */
mxArray *mock_dense(size_t m, size_t n, const double *data) {
  mxArray *a = mxCreateDoubleMatrix(m, n, mxREAL);
  if (data) memcpy(a->pr, data, m * n * sizeof(double));
  return a;
}
mxArray *mock_sparse(size_t m, size_t n, const size_t *jc, const size_t *ir, const double *pr) { /* copies: Matlab owns its arrays */
  mxArray *a = new_arr(K_SPARSE, m, n);
  const size_t nnz = jc[n];
  a->jc = (mwIndex *)mxMalloc((n + 1) * sizeof(mwIndex)); memcpy(a->jc, jc, (n + 1) * sizeof(mwIndex));
  a->ir = (mwIndex *)mxMalloc((nnz ? nnz : 1) * sizeof(mwIndex)); memcpy(a->ir, ir, nnz * sizeof(mwIndex));
  a->pr = (double *)mxMalloc((nnz ? nnz : 1) * sizeof(double)); memcpy(a->pr, pr, nnz * sizeof(double));
  return a;
}
mxArray *mock_struct(void) { return mxCreateStructMatrix(1, 1, 0, NULL); }
void mock_struct_add(mxArray *s, const char *name, mxArray *v) {
  s->names = (char **)realloc(s->names, (s->nfields + 1) * sizeof(char *));
  s->vals = (mxArray **)realloc(s->vals, (s->nfields + 1) * sizeof(mxArray *));
  s->names[s->nfields] = strdup(name); s->vals[s->nfields] = v; s->nfields++;
}
/* run mexFunction(nlhs, plhs, 2, {data, settings}); 0 on return, 1 when the gateway raised mexErrMsgTxt (text in mock_last_error) */
int mock_call(int nlhs, mxArray **plhs, const mxArray *data, const mxArray *settings) {
  const mxArray *prhs[2] = {data, settings};
  g_err[0] = 0;
  if (setjmp(g_jmp)) { g_jmp_armed = 0; return 1; }
  g_jmp_armed = 1;
  mexFunction(nlhs, plhs, 2, prhs);
  g_jmp_armed = 0;
  return 0;
}
/* the same for a gateway of three inputs ([sol, info] = abip_qcp(data, cones, settings)); the gateway checks nrhs itself */
int mock_calln(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs) {
  g_err[0] = 0;
  if (setjmp(g_jmp)) { g_jmp_armed = 0; return 1; }
  g_jmp_armed = 1;
  mexFunction(nlhs, plhs, nrhs, prhs);
  g_jmp_armed = 0;
  return 0;
}
const char *mock_last_error(void) { return g_err; }
size_t mock_numel(const mxArray *a) { return mxGetNumberOfElements(a); }
const double *mock_data(const mxArray *a) { return a->pr; }
const char *mock_string(const mxArray *a) { return a && a->kind == K_STRING ? a->str : NULL; }
mxArray *mock_field(const mxArray *s, const char *name) { return mxGetField(s, 0, name); }
int mock_nfields(const mxArray *s) { return s && s->kind == K_STRUCT ? s->nfields : -1; }
const char *mock_field_name(const mxArray *s, int f) { return s->names[f]; }
