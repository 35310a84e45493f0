// calc_cost_sgm MEX gateway -- drop-in for the reference's calc_cost_sgm.cpp:539-598.
//   [bestD, minC, conf, bestD2] = calc_cost_sgm(I1, I2, dMax, vMax, pixelPosD0, normlizeDirection,
//                                               offsetFromPosD0, P1, P2)
// called as [bestD, minC] = ... from epipolar_sgm_of.m:45.  Everything computes on the GPU through
// libfsgm_hip.so; this file only unpacks mxArrays.
// Environment: FSGM_DEVICE (HIP ordinal, default 0); FSGM_DEVICES (a list, for batches: below); FSGM_EPI_PATHS=8 enables the diagonal paths
// the reference compiles out (calc_cost_sgm.cpp:104) -- default 4 = as shipped; FSGM_EPI_FB_CHECK=1
// runs the forward-backward check the reference has commented out (:589-590) and fills conf/bestD2.
#include "gateway_common.h"
#include <vector>

// A BATCH (extension; a 2-D I1 is the reference's call and behaves as above): I1, I2 uint8 W x H x n, pixelPosD0 and
// normlizeDirection double W x H x 2 x n, offsetFromPosD0 double W x H x n  ->  bestD, minC (conf, bestD2) W x H x n.
// The frames are independent calls of the reference MEX; they run together on the GPU, and with FSGM_DEVICES=0,1,...,7 frame i
// runs on the i-th entry (mod the list length) -- one host thread per entry inside libfsgm_hip.so, no collective -- so one
// MATLAB process drives the 8 GPUs of a node.  FSGM_DEVICES unset: everything on FSGM_DEVICE.
static void batch_call(int nlhs, mxArray* plhs[], const mxArray* prhs[]) {
    const char* fn = "calc_cost_sgm";
    const mwSize* d = mxGetDimensions(prhs[0]);
    const size_t W = d[0], H = d[1], n = d[2];
    if (mxGetClassID(prhs[0]) != mxUINT8_CLASS || mxGetClassID(prhs[1]) != mxUINT8_CLASS || mxIsComplex(prhs[0]) || mxIsComplex(prhs[1]))
        mexErrMsgIdAndTxt("fsgm:class", "%s: I1 and I2 must be real uint8 arrays", fn);
    if (W == 0 || H == 0 || n == 0) mexErrMsgIdAndTxt("fsgm:size", "%s: I1 is empty", fn);
    if (mxGetNumberOfDimensions(prhs[1]) != 3 || mxGetNumberOfElements(prhs[1]) != W * H * n || mxGetM(prhs[1]) != W)
        mexErrMsgIdAndTxt("fsgm:size", "%s: I2 must be %zu x %zu x %zu like I1", fn, W, H, n);
    const uint8_t* I1 = (const uint8_t*)mxGetData(prhs[0]);
    const uint8_t* I2 = (const uint8_t*)mxGetData(prhs[1]);
    const int dMax = need_int(fn, prhs[2], "dMax");
    const double vMax = need_scalar(fn, prhs[3], "vMax");
    const double* pd0 = need_f64(fn, prhs[4], "pixelPosD0", W * H * 2 * n);
    const double* nd = need_f64(fn, prhs[5], "normlizeDirection", W * H * 2 * n);
    const double* off = need_f64(fn, prhs[6], "offsetFromPosD0", W * H * n);
    const int P1 = need_int(fn, prhs[7], "P1"), P2 = need_int(fn, prhs[8], "P2");
    if (dMax < 1) mexErrMsgIdAndTxt("fsgm:range", "%s: dMax must be >= 1", fn);
    mxArray* bestD = new_array(W, H, n, mxUINT32_CLASS);
    mxArray* minC = new_array(W, H, n, mxUINT32_CLASS);
    plhs[0] = bestD;
    if (nlhs > 1) plhs[1] = minC;
    if (nlhs > 2) plhs[2] = new_array(W, H, n, mxUINT8_CLASS);
    if (nlhs > 3) plhs[3] = new_array(W, H, n, mxUINT32_CLASS);
    std::vector<fsgm_epi_in> in(n);
    std::vector<fsgm_epi_out> out(n);
    const size_t NP = W * H;
    for (size_t i = 0; i < n; i++) {
        in[i].I1 = I1 + i * NP; in[i].I2 = I2 + i * NP; in[i].width = (int32_t)W; in[i].height = (int32_t)H;
        in[i].dMax = dMax; in[i].vMax = vMax; in[i].P1 = P1; in[i].P2 = P2;
        in[i].pixelPosD0 = pd0 + i * 2 * NP; in[i].normDir = nd + i * 2 * NP; in[i].offset = off + i * NP;
        out[i].bestD = (uint32_t*)mxGetData(bestD) + i * NP; out[i].minC = (uint32_t*)mxGetData(minC) + i * NP;
        out[i].C = NULL; out[i].S = NULL;
        out[i].conf = nlhs > 2 ? (uint8_t*)mxGetData(plhs[2]) + i * NP : NULL;
        out[i].bestD2 = nlhs > 3 ? (uint32_t*)mxGetData(plhs[3]) + i * NP : NULL;
    }
    fsgm_epi_params prm = fsgm_epi_params_default();
    prm.device = fsgm_env_int("FSGM_DEVICE", 0);
    prm.paths = fsgm_env_int("FSGM_EPI_PATHS", 4);
    prm.fb_check = fsgm_env_int("FSGM_EPI_FB_CHECK", 0) != 0;
    int32_t devs[64];
    const char* list = getenv("FSGM_DEVICES");
    int32_t nd_list = (list && *list) ? fsgm_parse_device_list(list, devs, 64) : 0;
    if (nd_list < 0) mexErrMsgIdAndTxt("fsgm:invalid", "%s: FSGM_DEVICES=\"%s\" is not a list of device ordinals", fn, list);
    if (nd_list == 0) { devs[0] = prm.device; nd_list = 1; }
    fsgm_register_atexit();
    const fsgm_status st = fsgm_calc_cost_sgm_batch_devices_host((int32_t)n, in.data(), out.data(), &prm, nd_list, devs);
    if (nlhs <= 1) mxDestroyArray(minC);
    check_status(fn, st);
}

extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const char* fn = "calc_cost_sgm";
    need_args(fn, nrhs, 9, nlhs, 4);
    if (mxGetNumberOfDimensions(prhs[0]) == 3) { batch_call(nlhs, plhs, prhs); return; }
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
