/* TEST INFRASTRUCTURE: the slice of Matlab's mex.h that mex/abip_hip_mex.c uses, so that the gateway can be compile-checked
 * (never linked or run) where no Matlab exists. */
#ifndef MOCK_MEX_H
#define MOCK_MEX_H
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;
mxArray *mxGetField(const mxArray *s, mwIndex i, const char *name);
double *mxGetPr(const mxArray *a);
int mxIsSparse(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
mwIndex *mxGetJc(const mxArray *a);
mwIndex *mxGetIr(const mxArray *a);
void *mxMalloc(size_t n);
void *mxCalloc(size_t n, size_t sz);
void mxFree(void *p);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateString(const char *s);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names);
void mxSetField(mxArray *s, mwIndex i, const char *name, mxArray *v);
void mxSetPr(mxArray *a, double *p);
void mxSetM(mxArray *a, mwSize m);
void mxSetN(mxArray *a, mwSize n);
int mexPrintf(const char *fmt, ...);
void mexErrMsgTxt(const char *msg);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#endif
