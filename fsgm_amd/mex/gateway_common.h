// gateway_common.h -- argument plumbing shared by the four mexFunction gateways.
// The reference validates nothing (a wrong class or size is a crash); the gateways check every
// argument and raise fsgm:* errors through mexErrMsgIdAndTxt (SURVEY 8b "errors").
#pragma once
#include "mex.h"
#include "fsgm.h"
#include <math.h>
#include <stdlib.h>

static bool g_atexit_registered = false;
static inline void fsgm_register_atexit() {
    if (!g_atexit_registered) { mexAtExit(fsgm_shutdown); g_atexit_registered = true; }
}

static inline int fsgm_env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

static inline void need_args(const char* fn, int nrhs, int want_rhs, int nlhs, int max_lhs) {
    if (nrhs != want_rhs) mexErrMsgIdAndTxt("fsgm:nrhs", "%s: %d inputs required, got %d", fn, want_rhs, nrhs);
    if (nlhs > max_lhs) mexErrMsgIdAndTxt("fsgm:nlhs", "%s: at most %d outputs, asked for %d", fn, max_lhs, nlhs);
}

static inline const uint8_t* need_u8_image(const char* fn, const mxArray* a, const char* name, size_t* W, size_t* H) {
    if (mxGetClassID(a) != mxUINT8_CLASS || mxIsComplex(a) || mxGetNumberOfDimensions(a) != 2)
        mexErrMsgIdAndTxt("fsgm:class", "%s: %s must be a real 2-D uint8 matrix (width x height after permute)", fn, name);
    if (*W == 0 && *H == 0) { *W = mxGetM(a); *H = mxGetN(a); }
    else if (mxGetM(a) != *W || mxGetN(a) != *H)
        mexErrMsgIdAndTxt("fsgm:size", "%s: %s must be %zu x %zu like I1", fn, name, *W, *H);
    if (*W == 0 || *H == 0) mexErrMsgIdAndTxt("fsgm:size", "%s: %s is empty", fn, name);
    return (const uint8_t*)mxGetData(a);
}

static inline const double* need_f64(const char* fn, const mxArray* a, const char* name, size_t numel) {
    if (mxGetClassID(a) != mxDOUBLE_CLASS || mxIsComplex(a))
        mexErrMsgIdAndTxt("fsgm:class", "%s: %s must be a real double array", fn, name);
    if (numel && mxGetNumberOfElements(a) != numel)
        mexErrMsgIdAndTxt("fsgm:size", "%s: %s must have %zu elements, has %zu", fn, name, numel, mxGetNumberOfElements(a));
    return mxGetPr(a);
}

static inline double need_scalar(const char* fn, const mxArray* a, const char* name) {
    if (mxGetNumberOfElements(a) != 1 || mxIsComplex(a))
        mexErrMsgIdAndTxt("fsgm:size", "%s: %s must be a real scalar", fn, name);
    return mxGetScalar(a);
}

// the reference assigns mxGetScalar() to int: C truncation
static inline int need_int(const char* fn, const mxArray* a, const char* name) {
    const double v = need_scalar(fn, a, name);
    if (!(v > -2147483648.0 && v < 2147483648.0)) mexErrMsgIdAndTxt("fsgm:range", "%s: %s out of int range", fn, name);
    return (int)v;
}

static inline int need_whole(const char* fn, const mxArray* a, const char* name, int lo) {
    const double v = need_scalar(fn, a, name);
    if (v != floor(v) || v < lo || v > 1e6) mexErrMsgIdAndTxt("fsgm:range", "%s: %s must be an integer >= %d", fn, name, lo);
    return (int)v;
}

static inline mxArray* new_array(size_t W, size_t H, size_t planes, mxClassID cls) {
    const mwSize dims[3] = {W, H, planes};
    return mxCreateNumericArray(planes > 1 ? 3 : 2, dims, cls, mxREAL);
}

static inline void check_status(const char* fn, fsgm_status st) {
    if (st == FSGM_OK) return;
    const char* id = st == FSGM_ERR_INVALID ? "fsgm:invalid" : st == FSGM_ERR_HIP ? "fsgm:hip"
                   : st == FSGM_ERR_NOMEM ? "fsgm:nomem" : "fsgm:unsupported";
    mexErrMsgIdAndTxt(id, "%s: %s", fn, fsgm_last_error());
}
