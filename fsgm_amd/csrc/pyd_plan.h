// pyd_plan.h -- the device-resident plan behind the calc_pyd_cost_sgm entry points, shared between
// capi_pyd.hip (one MEX call = one level) and capi_pyramid.hip (the pyramidal_sgm.m level loop,
// which writes images and hint maps straight into the level plans' HBM buffers).
#pragma once
#include "capi_common.h"
#include <vector>

struct fsgm_pyd_plan {
    int W = 0, H = 0, mvW = 0, mvH = 0, rX = 0, rY = 0, rAgg = 0, batch = 0, device = 0;
    int Sx = 0, Sy = 0, D = 0;
    int RS = 0, PS = 0;                  // volume layout in HBM (pyd_kernels.h): row stride, bytes per pixel
    int P1 = 6, P2 = 32, diagonal = 1, totalPass = 2, adaptive = 0, subpixel = 0;   // pyramidal_sgm.m:15-22
    int cmax = 24;                       // upper bound of the values in dC
    size_t NP = 0, N = 0, MV = 0;        // N = bytes of one volume (NP * PS)
    hipStream_t stream = nullptr;
    bool owns_stream = true;             // false when a pyramid plan lends its stream (capi_pyramid.hip)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t *dI1 = nullptr, *dI2 = nullptr, *dC = nullptr, *dL = nullptr;
    uint32_t *dCen1 = nullptr, *dCen2 = nullptr, *dBestD = nullptr, *dMinC = nullptr, *dS = nullptr, *dDesc = nullptr;
    std::vector<uint8_t> stage;          // host staging for layout conversion of debug volumes
    double *dMv = nullptr, *dMvSub = nullptr;
};

namespace fsgm {
// enqueue the FSGM_STAGE_* stages of one level on the plan's stream (no synchronisation);
// dS (may be null): debug tap for the summed path costs, u32 [batch][NP][D]
fsgm_status pyd_enqueue(fsgm_pyd_plan* p, int stages, uint32_t* dS);
}  // namespace fsgm
