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
    uint32_t* K4;           // [frames][NP][D] 4-byte entries (ng_key4) -- the 3x3 hint kernel's output when every key fits; null: 12-byte entries only
    uint32_t* flags;        // two words zeroed by the caller, = { unsafe, a key that does not fit 4 bytes }; needed with K4
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
    const uint32_t* ck;     // [frames][NP][D] packed motion vector (ng_pack_mv) of the kept entry at each place   } launch_ng_dedupe; the
    const uint16_t* cm;     // [frames][NP][D] (index of the group's first member << 8) | cost of the kept entry     } compact kernel's input
    int role;               // which of the launch's aggregation kernels this is (NG_ROLE_*): each decides on the device whether it runs
    int with_compact;       // the compact kernel is part of this launch set
    int compact_g;          // compact kernel: lanes a line of this launch (16 / 32 / 64); the one the list statistics favour runs
    int compact_force;      // ... or this one whatever they say (FSGM_NG_COMPACT_G, tests)
    int blk_begin_c[5];     // compact kernel: first block of each range (4 lines a workgroup)
    int slot_of_c[4];
    int16_t* L4;            // [frames][NP][4 path slots][64] the compact kernel's path costs of the kept entries, by place in the pixel's list:
                            // plain 2-byte stores, contiguous per pixel and path, instead of one atomic add to S per kept entry and path
                            // (208 M scattered atomics per batch of 8 at 1242x375: 1.6 of 8.3 ms).  The WTA sums the four slots.
                            // Null: the compact kernel is not launched
    uint32_t* kstat;        // [256] partial sums of the list lengths of a sample of this launch's pixels, [256] flags: bit 0 a list longer
                            // than 64 entries, bit 1 a pixel whose entries do not fit the packed key (launch_ng_dedupe); [257] set by the
                            // compact kernel when it is the one that runs: S holds nothing then, the sums are in L4
    int W, H, D;
    int P1, P2;
    int blk_begin[5];
    int slot_of[4];         // split kernel: path slot of block range k (long lines are launched first)
};

struct NgWtaArgs {
    const Cand* C;
    const uint32_t* S;
    const uint16_t* cm;     // launch_ng_dedupe's kept-entry table and list lengths: the search runs over the groups of repeats,
    const uint8_t* dk;      // whose sums sit at their first members' indices; both null: over all D candidates
    const int16_t* L4;      // the compact kernel's per-path costs and the launch's statistics words (kstat[257] != 0: sums = the four
    const uint32_t* kstat;  // slots of L4 at the entry's place; else S); null: S
    const uint32_t* K4;     // 4-byte entries and the cost kernel's flags (flags[1] == 0: every key fits): the winner's motion vector comes
    const uint32_t* flags;  // from its key when the compact kernel ran; null: from C
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
enum { NG_ROLE_ANY = 0, NG_ROLE_GRID = 1, NG_ROLE_LIST = 2, NG_ROLE_COMPACT = 3, NG_ROLE_REST = 4 };
constexpr int NG_KSTAT_WORDS = 259;
constexpr int NG_L4_PER_PIXEL = 4 * 64;      // int16 entries of L4 per pixel
// kstat: NG_KSTAT_WORDS words, zeroed here; ck / cm may be null (no compact kernel)
void launch_ng_dedupe(hipStream_t st, const Cand* C, uint16_t* dd, uint8_t* dk, uint32_t* dbox, uint32_t* kstat, uint32_t* ck, uint16_t* cm, int W, int H, int D, int frames,
                      const uint32_t* K4 = nullptr, const uint32_t* flags = nullptr);
// With 4-byte entries (K4 in the memory of S): after the dedupe kernel and before launch_ng_aggregate -- decides on the device whether the
// compact matcher runs (kstat[258]); if not, expands the keys to the Cand list the general matchers read and zeroes S
void launch_ng_prepare_matchers(hipStream_t st, const NgAggArgs& a, const uint32_t* K4, Cand* C, const uint32_t* flags, int frames);
// S of the repeats := S of the entries they repeat (only needed when S itself is read back: the WTA looks them up)
void launch_ng_fill_repeats(hipStream_t st, uint32_t* S, const uint16_t* dd, const uint16_t* cm, int W, int H, int D, int frames);
// S at the kept entries' first-member indices := the sum of L4's four slots, when the compact kernel ran (kstat[257]): only for reading S back
void launch_ng_l4_to_s(hipStream_t st, uint32_t* S, const int16_t* L4, const uint16_t* cm, const uint8_t* dk, const uint32_t* kstat, int W, int H, int D, int frames);
void launch_ng_wta(hipStream_t st, const NgWtaArgs& a, int frames);
void launch_ng_subpixel(hipStream_t st, const NgSubpixArgs& a, int frames);
void launch_otf(hipStream_t st, const OtfArgs& a, int frames);

}  // namespace fsgm
