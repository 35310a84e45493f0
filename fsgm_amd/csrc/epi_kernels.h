// epi_kernels.h -- launch interface of the calc_cost_sgm-path kernels (epi_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define FSGM_GENERIC_MAX_D 1024

namespace fsgm {

struct EpiCostArgs {
    const uint32_t* cen1;   // [frames][NP]
    const uint32_t* cen2;
    const double* pd0;      // [frames][2][NP]
    const double* nd;       // [frames][2][NP]
    const double* off;      // [frames][NP]
    const double* vz;       // [D]  vzInd(d), tabulated on the host with the reference's expression
    double vzmax;           // max |vz[d]| (inf if any is not finite): bounds the sample positions per pixel
    uint8_t* Craw;          // [frames][NP][D]
    int W, H, D;
};

struct AggArgs {
    const uint8_t* C;       // [frames] cost volumes
    uint8_t* L;             // [frames][paths] per-path costs
    size_t c_frame_stride;  // bytes between frames of C
    size_t l_frame_stride;  // bytes between frames of L
    size_t l_dir_stride;    // bytes between path slots of L (= NP*D)
    int W, H, D;
    int P1, P2;
    int ndirs;              // path slots (4 or 8)
    int blk_begin[9];       // first block of each slot, [8] = total
    int dir_code[8];        // bits 1:0 = 0 along x, 1 along y, 2 (+1,+1), 3 (-1,+1); bit 2 = point-mirrored (pass 1)
};

struct WtaArgs {
    const uint8_t* L;
    size_t l_frame_stride, l_dir_stride;
    const double* off;      // [frames][NP]
    uint32_t* bestD;        // [frames][NP]
    uint32_t* minC;
    double vMax;
    int W, H, D;
    int ndirs;
    int subpixel, vz_to_disp;
};

// fused-sweep aggregation (epi_sweep.hip)
struct SweepArgs {
    const uint8_t* C;         // [frames] cost volumes
    size_t c_frame_stride;
    uint8_t* X;               // [frames] u8 sums of a sweep's three y (epi_sweep.hip): written by modes 0/1, read (the other sweep's) by modes 2/3
    size_t x_frame_stride;
    const uint8_t* Lh;        // final modes (2, 3): [frames] the horizontal pair's sum Y_h (pair kernels)
    size_t lh_frame_stride;
    int lh_natural;           // Lh is in natural d order (pairx_* kernels), not the sweeps' private one
    uint2* rec;               // mode 2: [frames][NP] {best | minC << 16, S[best-1] | S[best+1] << 16} (sums of u8 path costs: 16 bits)
    uint16_t* s0;             // mode 2: [frames][NP] S[0] of every pixel
    const uint8_t* state_in;  // [frames][3][W][D] normalised path states of the row above this block
    uint8_t* state_out;       // same, written for the next block
    size_t state_frame_stride;
    int W, H, D;
    int P1, P2;
    int y0, rows;             // rows [y0, y0+rows) of the sweep frame
};

// band sweeps: all four paths of a raster pass in one sweep, one workgroup per frame (epi_band.hip)
struct BandArgs {
    const uint8_t* C;         // [frames] cost volumes
    size_t c_frame_stride;
    uint8_t* Y;               // [frames] low bytes of the first pass's sum of y (private byte order): written by mode 0, read by mode 2
    size_t y_frame_stride;
    uint32_t* Yb;             // [frames][NP][LPP] its 9th bits, one dword per lane (8 paths with 4*P2 > 255), else unused
    size_t yb_frame_stride;   // in dwords
    uint4* edge;              // [frames][W][states][LPP] path states of a band's last row for the band below (3 states at 8 paths, 1 at 4)
    size_t edge_frame_stride; // in uint4
    uint2* rec;               // mode 2: [frames][NP] {best | minC << 16, S[best-1] | S[best+1] << 16} (sums of u8 path costs: 16 bits)
    uint16_t* s0;             // mode 2: [frames][NP] S[0] of every pixel
    uint32_t* Sdbg;           // mode 2, optional: natural-order u32 dump of S [frames][NP][D] (debug tap)
    // chained form: one workgroup per (band, frame), the bands of a frame handing over while they run
    int chain;                // 0: one workgroup per frame, band after band (edge: one map per frame, reused in place)
    uint32_t* ticket;         // work counter, zero at launch
    uint32_t tag;             // 4-bit launch sequence number spread over the top bits of a dword's bytes
    uint32_t* err;            // set to non-zero when a hand-off wait gave up
    int frames, nbands, group; // group: frames whose bands are dealt band-major (tickets)
    int W, H, D;
    int P1, P2;
};

struct SweepSumArgs {          // what wta_sweep_kernel adds up (u8 volumes; the Y volumes in the sweeps' private byte order)
    const uint8_t* C;
    const uint8_t* Xdn;       // Y of the down sweep (or of the vertical pair)
    const uint8_t* Xup;       // Y of the up sweep; may be null
    size_t v_frame_stride;
    const uint8_t* Lh;        // [frames] Y_h of the horizontal pair
    size_t lh_frame_stride;
    int lh_natural;           // Lh is in natural d order (pairx_* kernels), not the private one
    const uint8_t* Lx;        // instead of Lh (then null): [frames][2][N] the two along-x PATH volumes L of the line kernels, natural d order
    size_t lx_frame_stride;
    int nC;                   // S = nC*(C + bias) - (Xdn + Xup + Lh) [+ Lx[0] + Lx[1]: a path's L = (C + bias) - (y + P1)]
    int bias;                 // P2 + P1: the Y volumes hold y + P1 per path (epi_step.h, step_b)
    uint32_t* Sdbg;           // optional natural-order u32 dump of S [frames][NP][D]
};

struct PairArgs {              // an opposite pair of paths as one excess sum (epi_sweep.hip, pair kernels)
    const uint8_t* C;         // [frames] cost volumes
    size_t c_frame_stride;
    uint8_t* X;               // [frames] u8 out: y_fwd + y_bwd (epi_sweep.hip)              (not in the final pass)
    size_t x_frame_stride;
    uint8_t* ckpt;            // [frames][lines][ntiles-1][D] normalised backward states at the tile boundaries
    size_t ckpt_frame_stride;
    const uint8_t* Xother;    // final pass: [frames] the other axis' sum
    size_t xo_frame_stride;
    uint2* rec;               // final pass: [frames][NP] {best | minC << 16, S[best-1] | S[best+1] << 16}
    uint16_t* s0;             // final pass: [frames][NP] S[0] of every pixel
    int nC;                   // final pass: S = nC * (C + P2) - (Y + Yother)
    int xo_natural;           // final pass: Xother is in natural d order (written by the pairx_* kernels), not the private one
    int W, H, D;
    int P1, P2;
};

struct FbArgs {                // forward-backward check (calc_cost_sgm.cpp:429-536)
    const uint32_t* D1;       // [frames][NP] bestD before vz->disparity
    const double* pd0;        // [frames][2][NP]
    const double* nd;         // [frames][2][NP]
    const double* off;        // [frames][NP]
    uint32_t* D2enc;          // [frames][NP] scratch: max(D1 + 1) scattered, 0 = invalid
    uint32_t* D2;             // [frames][NP] out: bestD2 (512<<8 where invalid)
    uint8_t* conf;            // [frames][NP] out: 1 = consistent
    double vMax;
    int W, H, n, thr;
};

enum { AGG_PACKED_NOWRAP = 0, AGG_PACKED_WRAP = 1, AGG_GENERIC = 2, AGG_SWEEP = 3, AGG_PAIRS = 4, AGG_BAND = 5 };

int  agg_packed_lpp(int D);   // lanes per pixel of the packed kernels, 0 if D is not 16<<k, k<=4
void launch_census(hipStream_t st, const uint8_t* img, uint32_t* cen, int W, int H, int frames);
void launch_epi_cost(hipStream_t st, const EpiCostArgs& a, uint8_t* C, int frames);
// the cost stage as one kernel (epi_cost.hip): raw costs + 5x5 box mean, a.Craw unused
bool costbox_ok(int W, int H, int D);
void launch_epi_costbox(hipStream_t st, const EpiCostArgs& a, uint8_t* C, int frames);
int  costbox_selftest(hipStream_t st);      // 0: the mean's fp16 multiply is exact on this device; > 0: mismatches; < 0: could not run
void launch_aggregate(hipStream_t st, AggArgs a, int paths, int frames, int kernel_kind);
void launch_wta(hipStream_t st, const WtaArgs& a, int frames, bool packed);
int    sweep_rows_per_launch(int D);
size_t sweep_state_bytes(int W, int D);   // one state buffer of one frame
void launch_sweep(hipStream_t st, const SweepArgs& a, int frames, int mode, int tall = 0);
// rows [ybeg, yend) of the sweep frame; modes 2 / 3: final on the mirrored / the pass-0 frame; *parity carries the state buffer in use from one range of a sweep to the next
void launch_sweep_rows(hipStream_t st, const SweepArgs& a, int frames, int mode, int tall, int ybeg, int yend, int* parity);   // 0 down, 1 up, 2 up + fused WTA; tall: 8-wave workgroups (modes 0, 1)
size_t pair_ckpt_bytes(int W, int H, int D, int axis);      // per frame; axis 0 horizontal, 1 vertical
void launch_pair(hipStream_t st, const PairArgs& a, int frames, int axis, bool final_pass, int phase = 0);
bool pair_x_fine_ok(int D);                                                        // the along-x pair with 8 costs a lane exists for this D
void launch_pair_x_fine(hipStream_t st, const PairArgs& a, int frames);            // checkpoint + sum pass, Y in natural d order
size_t band_edge_uint4s(int W, int D, int paths);           // one hand-off map between two bands of one frame, in uint4
int    band_rows(int D);                                    // rows per band
size_t band_bits_u32s(int W, int H, int D);                 // bit plane of one frame, in dwords
bool   band_needs_bits(int paths, int P1, int P2);
bool   band_ok(int D, int paths, int P1, int P2, int cmax);
void launch_band(hipStream_t st, const BandArgs& a, int frames, int paths, int mode);   // 0 first pass, 2 second pass + WTA records
void launch_sweep_finish(hipStream_t st, const WtaArgs& a, const uint2* rec, const uint16_t* s0, int frames);
void launch_wta_sweep(hipStream_t st, const WtaArgs& a, const SweepSumArgs& q, int frames);
void launch_fb_check(hipStream_t st, const FbArgs& a, int frames);
void launch_vz_convert(hipStream_t st, uint32_t* bestD, const double* off, int W, int H, int D, double vMax, int frames);
int  fused_step_selftest(hipStream_t st);   // 0: pk_max3 / pk_min3 of the fused step are exact u16 operations here
void launch_copy16(hipStream_t st, void* dst, const void* src, size_t bytes);   // bytes % 4096 == 0; bandwidth probe
void launch_sum_paths(hipStream_t st, const uint8_t* L, uint32_t* S, size_t n, size_t dir_stride, int ndirs);

}  // namespace fsgm
