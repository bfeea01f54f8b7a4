/* TEST STUB -- not MathWorks' matrix.h.  Declarations of the few mx* functions cnmf_e_amd/csrc/matlab/cnmfe_mex.cpp uses, with the
 * signatures MATLAB documents for them, so that the gateway can be syntax- and type-checked in an image without MATLAB
 * (tests/test_host_logic.py::test_mex_gateway_compiles_against_stub).  Nothing links against this. */
#ifndef MEX_STUB_MATRIX_H
#define MEX_STUB_MATRIX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7, mxINT32_CLASS = 12 } mxClassID;
bool mxIsChar(const mxArray *); bool mxIsEmpty(const mxArray *); bool mxIsSparse(const mxArray *); bool mxIsDouble(const mxArray *);
bool mxIsSingle(const mxArray *); bool mxIsLogical(const mxArray *); bool mxIsInt32(const mxArray *); bool mxIsUint16(const mxArray *); bool mxIsUint8(const mxArray *);
int mxGetString(const mxArray *, char *, mwSize);
double mxGetScalar(const mxArray *);
double *mxGetPr(const mxArray *);
void *mxGetData(const mxArray *);
mxLogical *mxGetLogicals(const mxArray *);
mwIndex *mxGetJc(const mxArray *); mwIndex *mxGetIr(const mxArray *);
size_t mxGetM(const mxArray *); size_t mxGetN(const mxArray *); size_t mxGetNumberOfElements(const mxArray *);
const char *mxGetClassName(const mxArray *);
double mxGetNaN(void);
void *mxMalloc(size_t); void *mxCalloc(size_t, size_t); void mxFree(void *);
mxArray *mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray *mxCreateDoubleScalar(double);
mxArray *mxCreateLogicalMatrix(mwSize, mwSize);
mxArray *mxCreateLogicalScalar(bool);
mxArray *mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxArray *mxCreateSparse(mwSize, mwSize, mwSize, mxComplexity);
void mxDestroyArray(mxArray *);
mxArray *mxDuplicateArray(const mxArray *);
#ifdef __cplusplus
}
#endif
#endif
