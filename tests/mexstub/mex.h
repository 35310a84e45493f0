/*
 * mex.h -- TEST STUB of the small part of MATLAB's MEX C API that the gateways in
 * fsgm_amd/mex/ use.  It exists so the gateways can be compiled and exercised on boxes without
 * MATLAB (this image has none).  It is never used to build anything from the reference tree.
 * On a MATLAB box the gateways are compiled against MATLAB's own mex.h instead
 * (INTEGRATION.md).
 */
#ifndef FSGM_TEST_MEX_H
#define FSGM_TEST_MEX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef size_t mwSize;
typedef enum { mxUNKNOWN_CLASS = 0, mxDOUBLE_CLASS = 6, mxUINT8_CLASS = 9, mxUINT32_CLASS = 13 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef struct mxArray_tag mxArray;

mxArray*  mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag);
mxArray*  mxCreateDoubleScalar(double v);
void      mxDestroyArray(mxArray* a);
void*     mxGetData(const mxArray* a);
double*   mxGetPr(const mxArray* a);
double    mxGetScalar(const mxArray* a);
size_t    mxGetM(const mxArray* a);
size_t    mxGetN(const mxArray* a);              /* product of dims 2..end, like MATLAB */
mwSize    mxGetNumberOfDimensions(const mxArray* a);
const mwSize* mxGetDimensions(const mxArray* a);
size_t    mxGetNumberOfElements(const mxArray* a);
mxClassID mxGetClassID(const mxArray* a);
int       mxIsComplex(const mxArray* a);
int       mexPrintf(const char* fmt, ...);
void      mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);   /* does not return */
int       mexAtExit(void (*fn)(void));

/* stub-only helpers for the test harness */
typedef void (*mexstub_fn)(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
int         mexstub_call(mexstub_fn fn, int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);   /* 0 ok, 1 error raised */
const char* mexstub_last_error_id(void);
const char* mexstub_last_error_msg(void);
const char* mexstub_printed(void);
void        mexstub_run_atexit(void);

#ifdef __cplusplus
}
#endif
#endif
