// TEST INFRASTRUCTURE -- a minimal stand-in for the part of MATLAB's mx* / mex* runtime that cnmf_e_amd/csrc/matlab/cnmfe_mex.cpp calls (the functions
// tests/mex_stub/{matrix,mex}.h declare), so that the gateway can be LINKED and RUN against libcnmfe_hip.so in an image without MATLAB:
// tests/test_gpu_mex_gateway.py builds this file + the gateway into one shared object, hands it arrays through the mock_* constructors below and calls
// mexFunction through mock_call.  Column-major dense arrays of class double / single / int32 / logical / uint16 / uint8 / char, and double / logical sparse
// matrices (compressed columns, the layout mxGetJc / mxGetIr document).  mexErrMsgIdAndTxt throws (MATLAB long-jumps); mock_call catches and hands the message
// back.  mxMalloc memory is released when mock_call returns, as MATLAB does when a MEX function exits.  Nothing of the product links against this.
#include "mex.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>

enum { CLS_DOUBLE = 6, CLS_SINGLE = 7, CLS_INT32 = 12, CLS_LOGICAL = 3, CLS_CHAR = 4, CLS_UINT16 = 11, CLS_UINT8 = 9 };
struct mxArray_tag {
    int cls; bool sparse; size_t m, n; void *data; mwIndex *jc, *ir; size_t nzmax;
};
static size_t elsize(int cls) { return cls == CLS_DOUBLE ? 8 : cls == CLS_SINGLE || cls == CLS_INT32 ? 4 : cls == CLS_UINT16 || cls == CLS_CHAR ? 2 : 1; }
static std::vector<void *> g_scratch;                  // mxMalloc / mxCalloc of the call in flight
static void (*g_atexit)(void) = nullptr;
struct MexError { std::string msg; };

static mxArray *make(int cls, size_t m, size_t n) {
    mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
    a->cls = cls; a->m = m; a->n = n; a->data = calloc(m * n + 1, elsize(cls));
    return a;
}
extern "C" {
bool mxIsChar(const mxArray *a) { return a->cls == CLS_CHAR; }
bool mxIsEmpty(const mxArray *a) { return a->m == 0 || a->n == 0; }
bool mxIsSparse(const mxArray *a) { return a->sparse; }
bool mxIsDouble(const mxArray *a) { return a->cls == CLS_DOUBLE; }
bool mxIsSingle(const mxArray *a) { return a->cls == CLS_SINGLE; }
bool mxIsLogical(const mxArray *a) { return a->cls == CLS_LOGICAL; }
bool mxIsInt32(const mxArray *a) { return a->cls == CLS_INT32; }
bool mxIsUint16(const mxArray *a) { return a->cls == CLS_UINT16; }
bool mxIsUint8(const mxArray *a) { return a->cls == CLS_UINT8; }
int mxGetString(const mxArray *a, char *buf, mwSize len) {
    if (a->cls != CLS_CHAR) return 1;
    const size_t n = a->m * a->n;
    if (n + 1 > len) return 1;
    for (size_t i = 0; i < n; ++i) buf[i] = (char)((const uint16_t *)a->data)[i];
    buf[n] = 0;
    return 0;
}
double mxGetScalar(const mxArray *a) {
    if (a->m * a->n == 0) return 0.0;
    switch (a->cls) {
    case CLS_DOUBLE: return ((const double *)a->data)[0];
    case CLS_SINGLE: return ((const float *)a->data)[0];
    case CLS_INT32: return ((const int32_t *)a->data)[0];
    case CLS_UINT16: case CLS_CHAR: return ((const uint16_t *)a->data)[0];
    default: return ((const uint8_t *)a->data)[0];
    }
}
double *mxGetPr(const mxArray *a) { return (double *)a->data; }
void *mxGetData(const mxArray *a) { return a->data; }
mxLogical *mxGetLogicals(const mxArray *a) { return (mxLogical *)a->data; }
mwIndex *mxGetJc(const mxArray *a) { return a->jc; }
mwIndex *mxGetIr(const mxArray *a) { return a->ir; }
size_t mxGetM(const mxArray *a) { return a->m; }
size_t mxGetN(const mxArray *a) { return a->n; }
size_t mxGetNumberOfElements(const mxArray *a) { return a->m * a->n; }
const char *mxGetClassName(const mxArray *a) {
    switch (a->cls) { case CLS_DOUBLE: return "double"; case CLS_SINGLE: return "single"; case CLS_INT32: return "int32"; case CLS_LOGICAL: return "logical";
                      case CLS_CHAR: return "char"; case CLS_UINT16: return "uint16"; default: return "uint8"; }
}
double mxGetNaN(void) { return NAN; }
void *mxMalloc(size_t n) { void *p = malloc(n ? n : 1); g_scratch.push_back(p); return p; }
void *mxCalloc(size_t n, size_t s) { void *p = calloc(n ? n : 1, s ? s : 1); g_scratch.push_back(p); return p; }
void mxFree(void *p) { for (auto &q : g_scratch) if (q == p) { free(p); q = nullptr; return; } }
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return make(CLS_DOUBLE, m, n); }
mxArray *mxCreateDoubleScalar(double v) { mxArray *a = make(CLS_DOUBLE, 1, 1); ((double *)a->data)[0] = v; return a; }
mxArray *mxCreateLogicalMatrix(mwSize m, mwSize n) { return make(CLS_LOGICAL, m, n); }
mxArray *mxCreateLogicalScalar(bool v) { mxArray *a = make(CLS_LOGICAL, 1, 1); ((mxLogical *)a->data)[0] = v; return a; }
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID c, mxComplexity) { return make((int)c, m, n); }
mxArray *mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity) {
    mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
    a->cls = CLS_DOUBLE; a->sparse = true; a->m = m; a->n = n; a->nzmax = nzmax ? nzmax : 1;
    a->data = calloc(a->nzmax, sizeof(double)); a->ir = (mwIndex *)calloc(a->nzmax, sizeof(mwIndex)); a->jc = (mwIndex *)calloc(n + 1, sizeof(mwIndex));
    return a;
}
void mxDestroyArray(mxArray *a) { if (!a) return; free(a->data); free(a->jc); free(a->ir); free(a); }
mxArray *mxDuplicateArray(const mxArray *s) {
    mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
    *a = *s;
    if (s->sparse) {
        a->data = malloc(s->nzmax * elsize(s->cls)); memcpy(a->data, s->data, s->nzmax * elsize(s->cls));
        a->ir = (mwIndex *)malloc(s->nzmax * sizeof(mwIndex)); memcpy(a->ir, s->ir, s->nzmax * sizeof(mwIndex));
        a->jc = (mwIndex *)malloc((s->n + 1) * sizeof(mwIndex)); memcpy(a->jc, s->jc, (s->n + 1) * sizeof(mwIndex));
    } else { a->data = malloc((s->m * s->n + 1) * elsize(s->cls)); memcpy(a->data, s->data, s->m * s->n * elsize(s->cls)); }
    return a;
}
void mexErrMsgIdAndTxt(const char *, const char *fmt, ...) {
    char buf[2048];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    throw MexError{buf};
}
int mexCallMATLAB(int nlhs, mxArray *plhs[], int nrhs, mxArray *prhs[], const char *name) {    // only sparse(full double matrix)
    if (strcmp(name, "sparse") || nlhs != 1 || nrhs != 1 || prhs[0]->sparse || prhs[0]->cls != CLS_DOUBLE) return 1;
    const mxArray *f = prhs[0];
    const double *v = (const double *)f->data;
    size_t nnz = 0;
    for (size_t i = 0; i < f->m * f->n; ++i) nnz += v[i] != 0.0;
    mxArray *s = mxCreateSparse(f->m, f->n, nnz, mxREAL);
    size_t e = 0;
    for (size_t j = 0; j < f->n; ++j) {
        s->jc[j] = e;
        for (size_t i = 0; i < f->m; ++i) if (v[j * f->m + i] != 0.0) { s->ir[e] = i; ((double *)s->data)[e] = v[j * f->m + i]; ++e; }
    }
    s->jc[f->n] = e;
    plhs[0] = s;
    return 0;
}
int mexAtExit(void (*fn)(void)) { g_atexit = fn; return 0; }
void mexMakeArrayPersistent(mxArray *) {}

// ---- what the Python side of the test uses ----
mxArray *mock_dense(int cls, size_t m, size_t n, const void *src) { mxArray *a = make(cls, m, n); if (src) memcpy(a->data, src, m * n * elsize(cls)); return a; }
mxArray *mock_string(const char *s) { const size_t n = strlen(s); mxArray *a = make(CLS_CHAR, 1, n); for (size_t i = 0; i < n; ++i) ((uint16_t *)a->data)[i] = (uint16_t)s[i]; return a; }
mxArray *mock_sparse(size_t m, size_t n, const int64_t *jc, const int64_t *ir, const double *pr, int logical) {
    mxArray *a = mxCreateSparse(m, n, (size_t)jc[n], mxREAL);
    for (size_t j = 0; j <= n; ++j) a->jc[j] = (mwIndex)jc[j];
    for (int64_t e = 0; e < jc[n]; ++e) a->ir[e] = (mwIndex)ir[e];
    if (logical) { a->cls = CLS_LOGICAL; free(a->data); a->data = calloc(a->nzmax, 1); for (int64_t e = 0; e < jc[n]; ++e) ((mxLogical *)a->data)[e] = true; }
    else for (int64_t e = 0; e < jc[n]; ++e) ((double *)a->data)[e] = pr[e];
    return a;
}
int mock_class(const mxArray *a) { return a->cls; }
int mock_is_sparse(const mxArray *a) { return a->sparse; }
size_t mock_nnz(const mxArray *a) { return a->sparse ? (size_t)a->jc[a->n] : a->m * a->n; }
void mock_free(mxArray *a) { mxDestroyArray(a); }
// runs mexFunction; 0 = returned normally, 1 = mexErrMsgIdAndTxt (message in err)
int mock_call(int nout, mxArray **pout, int nin, mxArray **pin, char *err, int errlen) {
    int rc = 0;
    try { mexFunction(nout, pout, nin, (const mxArray **)pin); }
    catch (const MexError &e) { snprintf(err, (size_t)errlen, "%s", e.msg.c_str()); rc = 1; }
    for (void *p : g_scratch) free(p);
    g_scratch.clear();
    return rc;
}
void mock_exit(void) { if (g_atexit) g_atexit(); g_atexit = nullptr; }
}
