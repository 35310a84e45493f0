// pyd_kernels.h -- launch interface of the calc_pyd_cost_sgm / calc_pyd_cost_sgm_ng kernels
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define FSGM_PYD_MAX_D 1024      // candidates per pixel the 2-D kernels accept (Sx*Sy)
#define FSGM_PYD_MAX_SIDE 64     // and each side of the search window (2*half+1)

namespace fsgm {

struct PydCostArgs {
    const uint32_t* cen1;   // [frames][NP]
    const uint32_t* cen2;
    const double* mv;       // [frames][2][mvH*mvW]
    uint8_t* C;             // [frames][NP][D]
    int W, H, mvW, mvH;
    int rAgg, rX, rY;
};

struct PydAggArgs {
    const uint8_t* I1;      // [frames][NP]
    const uint8_t* C;       // [frames][NP][D]
    const double* mv;       // [frames][2][mvH*mvW]
    uint8_t* L;             // [frames][slots][NP][D]
    int W, H, mvW, mvH;
    int Sx, Sy;
    int P1, P2, adaptive;
    int ndirs;              // path slots
    int blk_begin[9];
    int dir_code[8];        // as AggArgs
};

struct PydWtaArgs {
    const uint8_t* L;
    uint32_t* bestD;        // [frames][NP]
    uint32_t* minC;
    double* mvSub;          // [frames][2][NP]
    uint32_t* S;            // optional debug tap [frames][NP][D] (may be null)
    int W, H, Sx, Sy;
    int ndirs;
    uint32_t weight[8];     // per slot: 1 for pass-0 paths, totalPass-1 for pass-1 paths
    int subpixel;
};

void launch_pyd_cost(hipStream_t st, const PydCostArgs& a, int frames);
// returns the number of path slots it planned (nd or 2*nd)
int  plan_pyd_dirs(PydAggArgs& a, int diagonal, int totalPass, uint32_t weight[8]);
// wrap = false: penalties in the no-wrap range (0 <= P1,P2, max C + P2 + max(P1,P2) <= 255)
void launch_pyd_aggregate(hipStream_t st, const PydAggArgs& a, int frames, bool wrap);
void launch_pyd_wta(hipStream_t st, const PydWtaArgs& a, int frames);

}  // namespace fsgm
