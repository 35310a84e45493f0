// ng_kernels.h -- launch interface of the neighbour-guided variants
// (calc_pyd_cost_sgm_ng.cpp and calc_cost_sgm_ng.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define FSGM_NG_MAX_D 512        // candidates per pixel of the hint-map variant: 9*(2r+1)^2, r <= 3

namespace fsgm {

struct Cand { int32_t mvx, mvy, cost; };     // calc_pyd_cost_sgm_ng.cpp:32-37

struct NgCostArgs {
    const uint32_t* cen1;   // [frames][NP]
    const uint32_t* cen2;
    const double* mv;       // [frames][2][mvH*mvW]
    Cand* C;                // [frames][NP][D]
    uint32_t* unsafe;       // one word (may be null), zeroed by the caller: set when a motion vector has |v| >= 2^30
    int W, H, mvW, mvH;
    int rAgg, rX, rY;
};

struct NgAggArgs {
    const Cand* C;          // [frames][NP][D]
    uint32_t* S;            // [frames][NP][D], zeroed before the launch; paths add atomically
    const uint32_t* unsafe; // the cost kernel's flag (null: take the generic kernel)
    const uint16_t* dd;     // [frames][NP][D] place of a candidate in its pixel's list without repeats, 0xFFFF = a repeat
    const uint8_t* dk;      //                 (launch_ng_dedupe); [frames][NP] length of that list.  Null: every candidate is staged
    const uint32_t* dbox;   // [frames][NP] packed origin of the bounding box of a pixel's motion vectors, or all ones when it is larger
                            // than the grid matcher takes (launch_ng_dedupe); null: list matcher only
    const uint32_t* kstat;  // [256] partial sums of the list lengths of a sample of this launch's pixels (launch_ng_dedupe); with `pick` != 0 a kernel
    int pick;               // leaves at once unless the mean length is >= NG_GRID_MIN_K (pick = 1) or below it (pick = -1)
    int W, H, D;
    int P1, P2;
    int blk_begin[5];
    int slot_of[4];         // split kernel: path slot of block range k (long lines are launched first)
};

struct NgWtaArgs {
    const Cand* C;
    const uint32_t* S;
    uint32_t* minC;         // [frames][NP]
    double* flow;           // [frames][2][NP]
    int W, H, D;
};

struct NgSubpixArgs {
    const uint32_t* cen1;
    const uint32_t* cen2;
    double* flow;
    int W, H;
};

// on-the-fly variant (calc_cost_sgm_ng.cpp): one workgroup walks one frame in raster order
constexpr int OTF_D = 108;       // DIRECTION_NUM*(N+M)*MV_PER_HINT  (:194)
constexpr int OTF_E = 110;       // + N best entries                  (:196)
struct OtfArgs {
    const uint8_t* I1;      // [frames][NP]
    const uint32_t* cen1;
    const uint32_t* cen2;
    const int32_t* rnd;     // [frames][NP*8]  libc rand() stream, 2 draws per random hint
    Cand* Lrow;             // [frames][3][2][W][OTF_E]  L2, L3, L4 double row buffers (zeroed)
    uint32_t* minC;         // [frames][NP]
    double* flow;           // [frames][2][NP]
    int W, H, P1, P2;
    int exact;              // 1: start in the exact matcher (FSGM_OTF_EXACT=1; otherwise entered when a motion vector leaves the packed range)
};

void launch_ng_cost(hipStream_t st, const NgCostArgs& a, int frames);
void launch_ng_aggregate(hipStream_t st, NgAggArgs a, int frames);
// repeats among the D <= 128 candidates of every pixel (same motion vector and same cost), see ng_dedupe_kernel
void launch_ng_dedupe(hipStream_t st, const Cand* C, uint16_t* dd, uint8_t* dk, uint32_t* dbox, uint32_t* kstat, int W, int H, int D, int frames);   // kstat: 256 words, zeroed here
void launch_ng_wta(hipStream_t st, const NgWtaArgs& a, int frames);
void launch_ng_subpixel(hipStream_t st, const NgSubpixArgs& a, int frames);
void launch_otf(hipStream_t st, const OtfArgs& a, int frames);

}  // namespace fsgm
