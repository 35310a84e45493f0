// calc_pyd_cost_sgm MEX gateway -- drop-in for the reference's calc_pyd_cost_sgm.cpp:439-510.
//   [bestD, minC, mvSub] = calc_pyd_cost_sgm(I1, I2, preMv, halfSearchWinSizeX, halfSearchWinSizeY,
//        aggHalfWinSize, subPixelRefine, P1, P2, enableDiagnalPath, totalPass, adpativeP2)
// called from pyramidal_sgm.m:50.
#include "gateway_common.h"

extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* fn = "calc_pyd_cost_sgm";
    need_args(fn, nrhs, 12, nlhs, 3);
    size_t W = 0, H = 0;
    fsgm_pyd_in in;
    in.I1 = need_u8_image(fn, prhs[0], "I1", &W, &H);
    in.I2 = need_u8_image(fn, prhs[1], "I2", &W, &H);
    in.width = (int32_t)W; in.height = (int32_t)H;
    in.preMv = need_f64(fn, prhs[2], "preMv", 0);
    in.mvWidth = (int32_t)mxGetM(prhs[2]);                            // :493
    in.mvHeight = (int32_t)(mxGetN(prhs[2]) / 2);                     // :494
    if (mxGetN(prhs[2]) % 2 || (size_t)in.mvWidth < W || (size_t)in.mvHeight < H)
        mexErrMsgIdAndTxt("fsgm:size", "%s: preMv must be mvW x mvH x 2 with mvW >= %zu and mvH >= %zu", fn, W, H);
    in.halfSearchWinSizeX = need_whole(fn, prhs[3], "halfSearchWinSizeX", 0);   // :457-459
    in.halfSearchWinSizeY = need_whole(fn, prhs[4], "halfSearchWinSizeY", 0);
    in.aggHalfWinSize = need_whole(fn, prhs[5], "aggHalfWinSize", 0);
    in.subPixelRefine = need_int(fn, prhs[6], "subPixelRefine");
    in.P1 = need_int(fn, prhs[7], "P1");
    in.P2 = need_int(fn, prhs[8], "P2");
    in.enableDiagnalPath = need_scalar(fn, prhs[9], "enableDiagnalPath") != 0;  // :464 bool
    in.totalPass = need_int(fn, prhs[10], "totalPass");
    in.adpativeP2 = need_scalar(fn, prhs[11], "adpativeP2") != 0;               // :466 bool
    const int dMax = (2 * in.halfSearchWinSizeX + 1) * (2 * in.halfSearchWinSizeY + 1);
    mexPrintf("width: %d, height: %d, dMax: %d, winRadiusAgg: %d\n", (int)W, (int)H, dMax, in.aggHalfWinSize);   // :491

    mxArray* bestD = new_array(W, H, 1, mxUINT32_CLASS);              // :474-476
    mxArray* minC = new_array(W, H, 1, mxUINT32_CLASS);
    mxArray* mvSub = new_array(W, H, 2, mxDOUBLE_CLASS);
    plhs[0] = bestD;
    if (nlhs > 1) plhs[1] = minC;
    if (nlhs > 2) plhs[2] = mvSub;
    fsgm_pyd_out out;
    out.bestD = (uint32_t*)mxGetData(bestD);
    out.minC = (uint32_t*)mxGetData(minC);
    out.mvSub = mxGetPr(mvSub);
    out.C = NULL; out.S = NULL;
    fsgm_register_atexit();
    const fsgm_status st = fsgm_calc_pyd_cost_sgm_host(&in, &out, fsgm_env_int("FSGM_DEVICE", 0));
    if (nlhs <= 1) mxDestroyArray(minC);
    if (nlhs <= 2) mxDestroyArray(mvSub);
    check_status(fn, st);
}
