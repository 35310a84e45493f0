// calc_cost_sgm MEX gateway -- drop-in for the reference's calc_cost_sgm.cpp:539-598.
//   [bestD, minC, conf, bestD2] = calc_cost_sgm(I1, I2, dMax, vMax, pixelPosD0, normlizeDirection,
//                                               offsetFromPosD0, P1, P2)
// called as [bestD, minC] = ... from epipolar_sgm_of.m:45.  Everything computes on the GPU through
// libfsgm_hip.so; this file only unpacks mxArrays.
// Environment: FSGM_DEVICE (HIP ordinal, default 0); FSGM_EPI_PATHS=8 enables the diagonal paths
// the reference compiles out (calc_cost_sgm.cpp:104) -- default 4 = as shipped; FSGM_EPI_FB_CHECK=1
// runs the forward-backward check the reference has commented out (:589-590) and fills conf/bestD2.
#include "gateway_common.h"

extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* fn = "calc_cost_sgm";
    need_args(fn, nrhs, 9, nlhs, 4);
    size_t W = 0, H = 0;
    fsgm_epi_in in;
    in.I1 = need_u8_image(fn, prhs[0], "I1", &W, &H);                 // :548, :562-563
    in.I2 = need_u8_image(fn, prhs[1], "I2", &W, &H);
    in.width = (int32_t)W; in.height = (int32_t)H;
    in.dMax = need_int(fn, prhs[2], "dMax");                          // :551
    in.vMax = need_scalar(fn, prhs[3], "vMax");                       // :552
    in.pixelPosD0 = need_f64(fn, prhs[4], "pixelPosD0", W * H * 2);   // :553
    in.normDir = need_f64(fn, prhs[5], "normlizeDirection", W * H * 2);
    in.offset = need_f64(fn, prhs[6], "offsetFromPosD0", W * H);
    in.P1 = need_int(fn, prhs[7], "P1");                              // :557-558
    in.P2 = need_int(fn, prhs[8], "P2");
    if (in.dMax < 1) mexErrMsgIdAndTxt("fsgm:range", "%s: dMax must be >= 1", fn);

    // outputs as the reference creates them (:569-572); conf and bestD2 stay zero because the
    // forward-backward check is commented out there (:589-590)
    mxArray* bestD = new_array(W, H, 1, mxUINT32_CLASS);
    mxArray* minC = new_array(W, H, 1, mxUINT32_CLASS);
    plhs[0] = bestD;
    if (nlhs > 1) plhs[1] = minC;
    if (nlhs > 2) plhs[2] = new_array(W, H, 1, mxUINT8_CLASS);
    if (nlhs > 3) plhs[3] = new_array(W, H, 1, mxUINT32_CLASS);

    fsgm_epi_out out;
    out.bestD = (uint32_t*)mxGetData(bestD);
    out.minC = (uint32_t*)mxGetData(minC);
    out.C = NULL; out.S = NULL;
    out.conf = nlhs > 2 ? (uint8_t*)mxGetData(plhs[2]) : NULL;
    out.bestD2 = nlhs > 3 ? (uint32_t*)mxGetData(plhs[3]) : NULL;
    fsgm_epi_params prm = fsgm_epi_params_default();
    prm.device = fsgm_env_int("FSGM_DEVICE", 0);
    prm.paths = fsgm_env_int("FSGM_EPI_PATHS", 4);
    prm.fb_check = fsgm_env_int("FSGM_EPI_FB_CHECK", 0) != 0;
    fsgm_register_atexit();
    const fsgm_status st = fsgm_calc_cost_sgm_host(&in, &out, &prm);
    if (nlhs <= 1) mxDestroyArray(minC);
    check_status(fn, st);
}
