/* mexstub.c -- implementation of the test stub declared in tests/mexstub/mex.h */
#include "mex.h"
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct mxArray_tag {
    mxClassID cls;
    mwSize ndim;
    mwSize dims[4];
    void* data;
};

static size_t elem_size(mxClassID c) { return c == mxDOUBLE_CLASS ? 8 : (c == mxUINT32_CLASS ? 4 : 1); }

static jmp_buf g_jmp;
static int g_in_call = 0;
static char g_err_id[128], g_err_msg[1024], g_printed[4096];
static void (*g_atexit[8])(void);
static int g_n_atexit = 0;

mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag) {
    (void)flag;
    mxArray* a = (mxArray*)calloc(1, sizeof(mxArray));
    a->cls = classid;
    a->ndim = ndim < 2 ? 2 : ndim;
    size_t n = 1;
    for (mwSize i = 0; i < a->ndim && i < 4; i++) {
        a->dims[i] = i < ndim ? dims[i] : 1;
        n *= a->dims[i];
    }
    a->data = calloc(n ? n : 1, elem_size(classid));          /* zero-initialised like MATLAB */
    return a;
}
mxArray* mxCreateDoubleScalar(double v) {
    const mwSize d[2] = {1, 1};
    mxArray* a = mxCreateNumericArray(2, d, mxDOUBLE_CLASS, mxREAL);
    *(double*)a->data = v;
    return a;
}
void mxDestroyArray(mxArray* a) { if (a) { free(a->data); free(a); } }
void* mxGetData(const mxArray* a) { return a->data; }
double* mxGetPr(const mxArray* a) { return (double*)a->data; }
double mxGetScalar(const mxArray* a) {
    switch (a->cls) {
        case mxDOUBLE_CLASS: return *(double*)a->data;
        case mxUINT32_CLASS: return *(uint32_t*)a->data;
        default: return *(uint8_t*)a->data;
    }
}
size_t mxGetM(const mxArray* a) { return a->dims[0]; }
size_t mxGetN(const mxArray* a) { size_t n = 1; for (mwSize i = 1; i < a->ndim; i++) n *= a->dims[i]; return n; }
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->ndim; }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims; }
size_t mxGetNumberOfElements(const mxArray* a) { size_t n = 1; for (mwSize i = 0; i < a->ndim; i++) n *= a->dims[i]; return n; }
mxClassID mxGetClassID(const mxArray* a) { return a->cls; }
int mxIsComplex(const mxArray* a) { (void)a; return 0; }

int mexPrintf(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    size_t used = strlen(g_printed);
    int n = vsnprintf(g_printed + used, sizeof(g_printed) - used, fmt, ap);
    va_end(ap);
    return n;
}
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    snprintf(g_err_id, sizeof(g_err_id), "%s", id ? id : "");
    vsnprintf(g_err_msg, sizeof(g_err_msg), fmt, ap);
    va_end(ap);
    if (g_in_call) longjmp(g_jmp, 1);
    fprintf(stderr, "mexErrMsgIdAndTxt outside mexstub_call: %s: %s\n", g_err_id, g_err_msg);
    abort();
}
int mexAtExit(void (*fn)(void)) {
    for (int i = 0; i < g_n_atexit; i++) if (g_atexit[i] == fn) return 0;
    if (g_n_atexit < 8) g_atexit[g_n_atexit++] = fn;
    return 0;
}
int mexstub_call(mexstub_fn fn, int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    g_err_id[0] = g_err_msg[0] = g_printed[0] = 0;
    g_in_call = 1;
    if (setjmp(g_jmp)) { g_in_call = 0; return 1; }
    fn(nlhs, plhs, nrhs, prhs);
    g_in_call = 0;
    return 0;
}
const char* mexstub_last_error_id(void) { return g_err_id; }
const char* mexstub_last_error_msg(void) { return g_err_msg; }
const char* mexstub_printed(void) { return g_printed; }
void mexstub_run_atexit(void) { for (int i = 0; i < g_n_atexit; i++) g_atexit[i](); g_n_atexit = 0; }
