// calc_pyd_cost_sgm_ng MEX gateway -- drop-in for the reference's calc_pyd_cost_sgm_ng.cpp:448-523.
//   [minC, flow] = calc_pyd_cost_sgm_ng(I1, I2, preMv, halfSearchWinSize, aggSize, subPixelRefine, P1, P2)
// (nothing in the reference calls it; its argument list equals the call in ng_sgm.m:20)
#include "gateway_common.h"

extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* fn = "calc_pyd_cost_sgm_ng";
    need_args(fn, nrhs, 8, nlhs, 2);
    size_t W = 0, H = 0;
    fsgm_ng_in in;
    in.I1 = need_u8_image(fn, prhs[0], "I1", &W, &H);
    in.I2 = need_u8_image(fn, prhs[1], "I2", &W, &H);
    in.width = (int32_t)W; in.height = (int32_t)H;
    in.preMv = need_f64(fn, prhs[2], "preMv", 0);
    in.mvWidth = (int32_t)mxGetM(prhs[2]);                            // :501-502
    in.mvHeight = (int32_t)(mxGetN(prhs[2]) / 2);
    // mvHeight = floor(N/2) exactly like the reference, so the single-plane zeros(row,col) that
    // ng_sgm.m:17 builds is accepted too (its lower half then serves as the y plane)
    if (in.mvWidth < 1 || in.mvHeight < 1)
        mexErrMsgIdAndTxt("fsgm:size", "%s: preMv must be mvW x (2*mvH), at least 1 x 2", fn);
    in.halfSearchWinSize = need_int(fn, prhs[3], "halfSearchWinSize");   // :488-489 (int) truncation
    in.aggSize = need_int(fn, prhs[4], "aggSize");                       // :490 (int)aggSize/2
    in.subPixelRefine = need_int(fn, prhs[5], "subPixelRefine");
    in.P1 = need_int(fn, prhs[6], "P1");
    in.P2 = need_int(fn, prhs[7], "P2");
    if (in.halfSearchWinSize < 0 || in.aggSize < 0) mexErrMsgIdAndTxt("fsgm:range", "%s: window sizes must be >= 0", fn);
    const int cph = (2 * in.halfSearchWinSize + 1) * (2 * in.halfSearchWinSize + 1);
    mexPrintf("width: %d, height: %d, dMax: %d, winRadiusAgg: %d\n", (int)W, (int)H, 9 * cph, in.aggSize / 2);   // :499

    mxArray* minC = new_array(W, H, 1, mxUINT32_CLASS);               // :476-477
    mxArray* flow = new_array(W, H, 2, mxDOUBLE_CLASS);
    plhs[0] = minC;
    if (nlhs > 1) plhs[1] = flow;
    fsgm_ng_out out;
    out.minC = (uint32_t*)mxGetData(minC);
    out.flow = mxGetPr(flow);
    out.S = NULL;
    fsgm_register_atexit();
    const fsgm_status st = fsgm_calc_pyd_cost_sgm_ng_host(&in, &out, fsgm_env_int("FSGM_DEVICE", 0));
    if (nlhs <= 1) mxDestroyArray(flow);
    check_status(fn, st);
}
