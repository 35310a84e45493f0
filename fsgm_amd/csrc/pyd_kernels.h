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
    int RS, PS;             // volume layout: candidate (sx, sy) of pixel p at p*PS + sx*RS + sy
};

struct PydAggArgs {
    const uint8_t* I1;      // [frames][NP]
    const uint8_t* C;       // [frames][NP][D]
    const double* mv;       // [frames][2][mvH*mvW]
    uint8_t* L;             // [frames][slots][NP][D]
    uint32_t* desc;         // [frames][slots][NP] per-step descriptors of the row-packed kernel (pyd_rows.hip)
    uint32_t* dump;         // 16 bytes that absorb the stores of lanes without a candidate row
    int W, H, mvW, mvH;
    int Sx, Sy;
    int RS, PS;
    int P1, P2, adaptive;
    int ndirs;              // path slots
    int blk_begin[9];
    int dir_code[8];        // as AggArgs
    int wide_mask;          // bit s: slot s runs the one-line-per-wave mapping of the row-packed kernel
};

struct PydWtaArgs {
    const uint8_t* L;
    uint32_t* bestD;        // [frames][NP]
    uint32_t* minC;
    double* mvSub;          // [frames][2][NP]
    uint32_t* S;            // optional debug tap [frames][NP][D] (may be null)
    int W, H, Sx, Sy;
    int RS, PS;
    int ndirs;
    uint32_t weight[8];     // per slot: 1 for pass-0 paths, totalPass-1 for pass-1 paths
    int subpixel;
};

// Volume layouts.  "rows": each of the Sx candidate rows padded to RS = 4*ceil(Sy/4) bytes, so one
// lane moves one row as RS/4 dwords (search windows up to 11x11, the reference's);  "compact":
// RS = Sy, PS = Sx*Sy = D, the reference's own order (calc_pyd_cost_sgm.cpp:392-393), any window.
inline bool pyd_rows_layout(int Sx, int Sy) { return Sx <= 11 && Sy <= 11; }
inline int  pyd_row_stride(int Sx, int Sy) { return pyd_rows_layout(Sx, Sy) ? 4 * ((Sy + 3) / 4) : Sy; }

void launch_pyd_cost(hipStream_t st, const PydCostArgs& a, int frames);
// returns the number of path slots it planned (nd or 2*nd); lines_per_block = 4 (generic kernel) or 16
// (row-packed); wide_rows: the horizontal slots of the row-packed kernel take its one-line-per-wave
// mapping (4 lines per block)
int  plan_pyd_dirs(PydAggArgs& a, int diagonal, int totalPass, uint32_t weight[8], int lines_per_block, bool wide_rows = false);
// wrap = false: penalties in the no-wrap range (0 <= P1,P2, max C + P2 + max(P1,P2) <= 255)
void launch_pyd_aggregate(hipStream_t st, const PydAggArgs& a, int frames, bool wrap);
void launch_pyd_wta(hipStream_t st, const PydWtaArgs& a, int frames);

// ---- row-packed kernels (pyd_rows.hip): rows layout, no-wrap penalties ----
bool pyd_rows_cost_ok(const PydCostArgs& a);                 // rows layout and aggregation radius <= 2
bool pyd_rows_wta_ok(const PydWtaArgs& a);                   // rows layout and weighted sums fit u16
void launch_pyd_rows_cost(hipStream_t st, const PydCostArgs& a, int frames);
void launch_pyd_rows_desc(hipStream_t st, const PydAggArgs& a, int frames);
void launch_pyd_rows_aggregate(hipStream_t st, const PydAggArgs& a, int frames);
void launch_pyd_rows_wta(hipStream_t st, const PydWtaArgs& a, int frames);

}  // namespace fsgm
