// fsgm_pyramidal_sgm MEX gateway -- the whole of pyramidal_sgm.m:24-76 in one call.
// The reference has no MEX of this name: pyramidal_sgm.m is a MATLAB function that calls the
// calc_pyd_cost_sgm MEX once per level.  A drop-in pyramidal_sgm.m keeps that
// function's signature and forwards to this gateway:
//   [mv, minC, mvPyd1, ..., mvPydN] = fsgm_pyramidal_sgm(I0p, I1p, numPyd)
//     I0p, I1p  uint8, width x height (gray) or width x height x 3 (RGB): permute(I, [2 1 3]) of the
//               images, the same order pyramidal_sgm.m:44-45 hands calc_pyd_cost_sgm
//     numPyd    pyramid levels (pyramidal_sgm.m:6-13; optional, default 5)
//     mv        double width x height x 2, level-1 flow (permute back like pyramidal_sgm.m:53)
//     minC      uint32 width x height, level-1 minimum summed path cost
//     mvPydl    double W_l x H_l x 2, flow of level l (the function's mvPyd{l})
// P1, P2 and the window sizes are pyramidal_sgm.m:15-22's constants; FSGM_PYD_* environment
// variables override them for experiments.
#include "gateway_common.h"

static const uint8_t* need_u8_planes(const char* fn, const mxArray* a, const char* name, size_t* W, size_t* H, size_t* ch) {
    const mwSize nd = mxGetNumberOfDimensions(a);
    if (mxGetClassID(a) != mxUINT8_CLASS || mxIsComplex(a) || nd < 2 || nd > 3)
        mexErrMsgIdAndTxt("fsgm:class", "%s: %s must be a real uint8 array, width x height or width x height x 3", fn, name);
    const mwSize* d = mxGetDimensions(a);
    const size_t c = nd == 3 ? d[2] : 1;
    if (c != 1 && c != 3) mexErrMsgIdAndTxt("fsgm:size", "%s: %s must have 1 or 3 planes, has %zu", fn, name, c);
    if (*W == 0 && *H == 0) { *W = d[0]; *H = d[1]; *ch = c; }
    else if (d[0] != *W || d[1] != *H || c != *ch)
        mexErrMsgIdAndTxt("fsgm:size", "%s: %s must have the size of I0", fn, name);
    if (*W == 0 || *H == 0) mexErrMsgIdAndTxt("fsgm:size", "%s: %s is empty", fn, name);
    return (const uint8_t*)mxGetData(a);
}

extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* fn = "fsgm_pyramidal_sgm";
    if (nrhs < 2 || nrhs > 3) mexErrMsgIdAndTxt("fsgm:nrhs", "%s: 2 or 3 inputs required, got %d", fn, nrhs);
    fsgm_pyramid_params prm = fsgm_pyramid_params_default();
    if (nrhs == 3) prm.numPyd = need_whole(fn, prhs[2], "numPyd", 1);
    if (prm.numPyd > 16) mexErrMsgIdAndTxt("fsgm:range", "%s: numPyd must be at most 16", fn);
    if (nlhs > 2 + prm.numPyd) mexErrMsgIdAndTxt("fsgm:nlhs", "%s: at most %d outputs, asked for %d", fn, 2 + prm.numPyd, nlhs);
    size_t W = 0, H = 0, ch = 0;
    const uint8_t* I0 = need_u8_planes(fn, prhs[0], "I0", &W, &H, &ch);
    const uint8_t* I1 = need_u8_planes(fn, prhs[1], "I1", &W, &H, &ch);
    prm.P1 = fsgm_env_int("FSGM_PYD_P1", prm.P1);
    prm.P2 = fsgm_env_int("FSGM_PYD_P2", prm.P2);
    prm.aggHalfWinSize = fsgm_env_int("FSGM_PYD_AGG", prm.aggHalfWinSize);
    prm.verSearchHalfWinSize = fsgm_env_int("FSGM_PYD_VER", prm.verSearchHalfWinSize);
    prm.horSearchHalfWinSize = fsgm_env_int("FSGM_PYD_HOR", prm.horSearchHalfWinSize);
    prm.adaptiveP2 = fsgm_env_int("FSGM_PYD_ADAPTIVE_P2", prm.adaptiveP2);
    prm.device = fsgm_env_int("FSGM_DEVICE", 0);

    mxArray* mv = new_array(W, H, 2, mxDOUBLE_CLASS);
    mxArray* minC = new_array(W, H, 1, mxUINT32_CLASS);
    plhs[0] = mv;
    if (nlhs > 1) plhs[1] = minC;
    double* lv[16] = {0};
    size_t w = W, h = H;
    for (int l = 0; l < prm.numPyd; l++) {
        if (nlhs > 2 + l) { plhs[2 + l] = new_array(w, h, 2, mxDOUBLE_CLASS); lv[l] = mxGetPr(plhs[2 + l]); }
        w = (w + 1) / 2; h = (h + 1) / 2;                                        // impyramid: ceil(size/2)
    }
    fsgm_register_atexit();
    const fsgm_status st = fsgm_pyramidal_sgm_host(I0, I1, (int32_t)W, (int32_t)H, (int32_t)ch, &prm, mxGetPr(mv),
                                                   (uint32_t*)mxGetData(minC), lv);
    if (nlhs <= 1) mxDestroyArray(minC);
    check_status(fn, st);
}
