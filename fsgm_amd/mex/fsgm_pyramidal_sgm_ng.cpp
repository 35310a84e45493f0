// fsgm_pyramidal_sgm_ng MEX gateway -- pyramidal_sgm.m's level loop (:24-76) around the neighbour-guided
// matcher calc_pyd_cost_sgm_ng, in one call (BASELINE config 4).  The reference ships that MEX without a
// driver (it is a swap-in for calc_cost_sgm_ng in ng_sgm.m:20); a MATLAB function with pyramidal_sgm's
// signature forwards to this gateway:
//   [flow, minC, flowPyd1, ..., flowPydN] = fsgm_pyramidal_sgm_ng(I0p, I1p, numPyd)
//     I0p, I1p  uint8, width x height (gray) or width x height x 3 (RGB): permute(I, [2 1 3]) of the images
//     numPyd    pyramid levels (optional, default 3: test_psgm.m:33)
//     flow      double width x height x 2, level-1 flow
//     minC      uint32 width x height, level-1 minimum summed path cost
//     flowPydl  double W_l x H_l x 2, flow of level l
// halfSearchWinSize, aggSize, subPixelRefine, P1, P2 are ng_sgm.m:7-8,20's values; FSGM_NG_* environment
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
    const char* fn = "fsgm_pyramidal_sgm_ng";
    if (nrhs < 2 || nrhs > 3) mexErrMsgIdAndTxt("fsgm:nrhs", "%s: 2 or 3 inputs required, got %d", fn, nrhs);
    fsgm_ng_pyramid_params prm = fsgm_ng_pyramid_params_default();
    if (nrhs == 3) prm.numPyd = need_whole(fn, prhs[2], "numPyd", 1);
    if (prm.numPyd > 16) mexErrMsgIdAndTxt("fsgm:range", "%s: numPyd must be at most 16", fn);
    if (nlhs > 2 + prm.numPyd) mexErrMsgIdAndTxt("fsgm:nlhs", "%s: at most %d outputs, asked for %d", fn, 2 + prm.numPyd, nlhs);
    size_t W = 0, H = 0, ch = 0;
    const uint8_t* I0 = need_u8_planes(fn, prhs[0], "I0", &W, &H, &ch);
    const uint8_t* I1 = need_u8_planes(fn, prhs[1], "I1", &W, &H, &ch);
    prm.P1 = fsgm_env_int("FSGM_NG_P1", prm.P1);
    prm.P2 = fsgm_env_int("FSGM_NG_P2", prm.P2);
    prm.halfSearchWinSize = fsgm_env_int("FSGM_NG_HALF_SEARCH", prm.halfSearchWinSize);
    prm.aggSize = fsgm_env_int("FSGM_NG_AGG_SIZE", prm.aggSize);
    prm.subPixelRefine = fsgm_env_int("FSGM_NG_SUBPIXEL", prm.subPixelRefine);
    prm.device = fsgm_env_int("FSGM_DEVICE", 0);

    mxArray* flow = new_array(W, H, 2, mxDOUBLE_CLASS);
    mxArray* minC = new_array(W, H, 1, mxUINT32_CLASS);
    plhs[0] = flow;
    if (nlhs > 1) plhs[1] = minC;
    double* lv[16] = {0};
    size_t w = W, h = H;
    for (int l = 0; l < prm.numPyd; l++) {
        if (nlhs > 2 + l) { plhs[2 + l] = new_array(w, h, 2, mxDOUBLE_CLASS); lv[l] = mxGetPr(plhs[2 + l]); }
        w = (w + 1) / 2; h = (h + 1) / 2;                                        // impyramid: ceil(size/2)
    }
    fsgm_register_atexit();
    const fsgm_status st = fsgm_pyramidal_sgm_ng_host(I0, I1, (int32_t)W, (int32_t)H, (int32_t)ch, &prm, mxGetPr(flow),
                                                      (uint32_t*)mxGetData(minC), lv);
    if (nlhs <= 1) mxDestroyArray(minC);
    check_status(fn, st);
}
