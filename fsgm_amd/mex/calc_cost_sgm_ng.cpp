// calc_cost_sgm_ng MEX gateway -- drop-in for the reference's calc_cost_sgm_ng.cpp:484-526.
//   [minC, flow] = calc_cost_sgm_ng(I1, I2, preMv, halfSearchWinSize, aggSize, subPixelRefine, P1, P2)
// called from ng_sgm.m:20.  Like the reference, arguments 3..6 are read and ignored (:497-503),
// and the random hints come from libc rand() (process-global state, :148-149): the library draws
// them on the host in the reference's order.
#include "gateway_common.h"

extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* fn = "calc_cost_sgm_ng";
    need_args(fn, nrhs, 8, nlhs, 2);
    size_t W = 0, H = 0;
    fsgm_otf_in in;
    in.I1 = need_u8_image(fn, prhs[0], "I1", &W, &H);
    in.I2 = need_u8_image(fn, prhs[1], "I2", &W, &H);
    in.width = (int32_t)W; in.height = (int32_t)H;
    in.P1 = need_int(fn, prhs[6], "P1");                              // :505-506
    in.P2 = need_int(fn, prhs[7], "P2");
    in.rand_stream = NULL;
    mexPrintf("dMax : %d\n", 108);                                    // :195

    mxArray* minC = new_array(W, H, 1, mxUINT32_CLASS);               // :512-513
    mxArray* flow = new_array(W, H, 2, mxDOUBLE_CLASS);
    plhs[0] = minC;
    if (nlhs > 1) plhs[1] = flow;
    fsgm_otf_out out;
    out.minC = (uint32_t*)mxGetData(minC);
    out.flow = mxGetPr(flow);
    fsgm_register_atexit();
    const fsgm_status st = fsgm_calc_cost_sgm_ng_host(&in, &out, fsgm_env_int("FSGM_DEVICE", 0));
    if (nlhs <= 1) mxDestroyArray(flow);
    check_status(fn, st);
}
