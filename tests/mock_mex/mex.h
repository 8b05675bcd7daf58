/* TEST INFRASTRUCTURE: the slice of Matlab's mex.h / matrix.h that the gateways use (mex/abip_hip_mex.c, mex/abip_hip_qcp_mex.c and the reference's
 * src/abip-lp/mexfile/abip_mex.c), so that they can be compiled AND EXECUTED where no Matlab exists.  The functions are
 * implemented by mock_mex_runtime.c (a few plain-C containers); nothing here ships with the product. */
#ifndef MOCK_MEX_H
#define MOCK_MEX_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;
mxArray *mxGetField(const mxArray *s, mwIndex i, const char *name);
double *mxGetPr(const mxArray *a);
int mxIsSparse(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
mwSize mxGetNumberOfDimensions(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
int mxIsEmpty(const mxArray *a);
mwIndex *mxGetJc(const mxArray *a);
mwIndex *mxGetIr(const mxArray *a);
void *mxMalloc(size_t n);
void *mxCalloc(size_t n, size_t sz);
void *mxRealloc(void *p, size_t n);
void mxFree(void *p);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateString(const char *s);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names);
mxArray *mxCreateStructArray(mwSize ndim, const mwSize *dims, int nfields, const char **names);
void mxSetField(mxArray *s, mwIndex i, const char *name, mxArray *v);
void mxSetPr(mxArray *a, double *p);
void mxSetM(mxArray *a, mwSize m);
void mxSetN(mxArray *a, mwSize n);
int mexPrintf(const char *fmt, ...);
void mexErrMsgTxt(const char *msg);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
