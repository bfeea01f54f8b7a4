/* TEST STUB -- not MathWorks' mex.h (see matrix.h beside it). */
#ifndef MEX_STUB_MEX_H
#define MEX_STUB_MEX_H
#include "matrix.h"
#ifdef __cplusplus
extern "C" {
#endif
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) __attribute__((noreturn));
int mexCallMATLAB(int nlhs, mxArray *plhs[], int nrhs, mxArray *prhs[], const char *name);
int mexAtExit(void (*fn)(void));
void mexMakeArrayPersistent(mxArray *);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
