// epi_kernels.hip -- gfx950 kernels for the calc_cost_sgm path
// (reference: calc_cost_sgm.cpp + common.cpp; citations per kernel).
//
// Data layout in HBM (all per frame, frames strided):
//   images  u8  [H][W]            census u32 [H][W]
//   maps    f64 [2][H][W] / [H][W]
//   cost volume C and per-path costs L_r: u8 [H][W][D], d fastest -> one pixel's D costs are
//   D contiguous bytes (128 B at D=128 = one L2 line), read 16 B per lane.
//
// No MFMA anywhere: this is integer stencil / scan work bound by HBM and VALU issue.
#include "epi_kernels.h"
#include <algorithm>
#include <type_traits>
#include "fsgm_device.h"
#include "epi_wta_tail.h"
#include "epi_step.h"

namespace fsgm {

// =============================================================================================
// census 5x5  (common.cpp:3-27): code = sum over the 25 taps, row-major from (-2,-2), of
// (nbr >= ctr) << (25 - tap); replicate border.
// =============================================================================================
__global__ __launch_bounds__(256) void census5x5_kernel(const uint8_t* __restrict__ img,
                                                        uint32_t* __restrict__ cen, int W, int H) {
    const int NP = W * H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const uint8_t* im = img + (size_t)blockIdx.y * NP;
    const int y = p / W, x = p - y * W;
    const unsigned ctr = im[p];
    uint32_t code = 0;
#pragma unroll
    for (int oy = -2; oy <= 2; oy++) {
        const int y2 = clampi(y + oy, 0, H - 1);
#pragma unroll
        for (int ox = -2; ox <= 2; ox++) {
            const int x2 = clampi(x + ox, 0, W - 1);
            code = (code + (im[y2 * W + x2] >= ctr ? 1u : 0u)) << 1;
        }
    }
    cen[(size_t)blockIdx.y * NP + p] = code;
}

// The same transform, four pixels a thread (round 4; the one-pixel kernel above spends 213 instructions a pixel, most of them
// the clamped addresses of its 25 byte loads, and ran at VALU busy 1.03: 0.005 of the cost stage's 0.046 ms per frame).  A thread
// owns pixels x0 .. x0 + 3 of a row (x0 a multiple of 4) and loads the 12 bytes x0 - 4 .. x0 + 7 of each of the five rows as
// three dwords; a tap is one compare with an SDWA byte select and one add-with-carry: code = 2 code + [nbr >= ctr], the
// reference's (code + bit) << 1 with the last shift applied at the end.  Threads whose bytes would leave the image -- the two
// border rows, the first and the last columns -- take the clamped form above.
template <int BYTE>
__device__ __forceinline__ unsigned long long census_cmp(const uint32_t w, const uint32_t ctr) {      // lane mask of [byte BYTE of w >= ctr]
    unsigned long long m;
    if (BYTE == 0) asm("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:DWORD src1_sel:BYTE_0" : "=s"(m) : "v"(ctr), "v"(w));
    if (BYTE == 1) asm("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:DWORD src1_sel:BYTE_1" : "=s"(m) : "v"(ctr), "v"(w));
    if (BYTE == 2) asm("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:DWORD src1_sel:BYTE_2" : "=s"(m) : "v"(ctr), "v"(w));
    if (BYTE == 3) asm("v_cmp_le_u32_sdwa %0, %1, %2 src0_sel:DWORD src1_sel:BYTE_3" : "=s"(m) : "v"(ctr), "v"(w));
    return m;
}
__device__ __forceinline__ void census_acc(uint32_t& code, const unsigned long long m) {              // code = 2 code + bit
    asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(code) : "s"(m) : "vcc");
}
// one tap of each of the thread's four pixels: the four compares first, then the four accumulations -- a mask is read three
// instructions after it is written (back to back the pair needs a wait state: an s_nop per tap)
#define FSGM_CENSUS_TAPS(w0, b0, w1, b1, w2, b2, w3, b3)                                                                  \
    do {                                                                                                                    \
        const unsigned long long m0 = census_cmp<b0>(w0, c0), m1 = census_cmp<b1>(w1, c1), m2 = census_cmp<b2>(w2, c2),     \
                                 m3 = census_cmp<b3>(w3, c3);                                                               \
        census_acc(k0, m0); census_acc(k1, m1); census_acc(k2, m2); census_acc(k3, m3);                                     \
    } while (0)

__global__ __launch_bounds__(256) void census5x5_quad_kernel(const uint8_t* __restrict__ img, uint32_t* __restrict__ cen, int W, int H) {
    const int Wq = (W + 3) >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Wq * H) return;
    const size_t NP = (size_t)W * H;
    const uint8_t* im = img + blockIdx.y * NP;
    uint32_t* out = cen + blockIdx.y * NP;
    const int y = q / Wq, x0 = (q - y * Wq) * 4;
    if (y >= 2 && y <= H - 3 && x0 >= 4 && x0 + 7 <= W - 1) {
        const uint8_t* r0 = im + (size_t)y * W + x0;
        const uint32_t cw = *(const uint32_t*)r0;                           // (rows start at y W: dword loads at any byte address)
        const uint32_t c0 = cw & 0xFFu, c1 = (cw >> 8) & 0xFFu, c2 = (cw >> 16) & 0xFFu, c3 = cw >> 24;
        uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
#pragma unroll
        for (int oy = -2; oy <= 2; oy++) {
            const uint8_t* r = r0 + (ptrdiff_t)oy * W;
            const uint32_t L = *(const uint32_t*)(r - 4), M = oy == 0 ? cw : *(const uint32_t*)r, R = *(const uint32_t*)(r + 4);
            // pixel 0: columns x0 - 2 .. x0 + 2 = L.2 L.3 M.0 M.1 M.2; pixel 1: L.3 M.0 .. M.3; pixel 2: M.0 .. M.3 R.0; pixel 3: M.1 .. M.3 R.0 R.1
            FSGM_CENSUS_TAPS(L, 2, L, 3, M, 0, M, 1);                       // dx = -2
            FSGM_CENSUS_TAPS(L, 3, M, 0, M, 1, M, 2);                       // dx = -1
            FSGM_CENSUS_TAPS(M, 0, M, 1, M, 2, M, 3);                       // dx = 0
            FSGM_CENSUS_TAPS(M, 1, M, 2, M, 3, R, 0);                       // dx = +1
            FSGM_CENSUS_TAPS(M, 2, M, 3, R, 0, R, 1);                       // dx = +2
        }
        uint32_t* o = out + (size_t)y * W + x0;
        o[0] = k0 << 1; o[1] = k1 << 1; o[2] = k2 << 1; o[3] = k3 << 1;
        return;
    }
    for (int i = 0; i < 4; i++) {                                             // borders: replicate (common.cpp:17-18)
        const int x = x0 + i;
        if (x >= W) break;
        const unsigned ctr = im[(size_t)y * W + x];
        uint32_t code = 0;
        for (int oy = -2; oy <= 2; oy++) {
            const int y2 = clampi(y + oy, 0, H - 1);
            for (int ox = -2; ox <= 2; ox++) code = (code + (im[(size_t)y2 * W + clampi(x + ox, 0, W - 1)] >= ctr ? 1u : 0u)) << 1;
        }
        out[(size_t)y * W + x] = code;
    }
}

// =============================================================================================
// raw Hamming cost along the epipolar line  (calc_cost_sgm.cpp:343-381).
// One thread = one pixel x 4 consecutive d.  fp64 geometry in the reference's association
// order with explicit round-to-nearest mul/add (no FMA contraction).
// =============================================================================================
template <bool VEC4>                                       // VEC4: D % 4 == 0, no per-element guards, one u32 store
__global__ __launch_bounds__(256) void epi_rawcost_kernel(EpiCostArgs a) {
    const int W = a.W, H = a.H, D = a.D;
    const int NP = W * H;
    const int Dq = (D + 3) >> 2;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;        // NP * Dq < 2^31 (checked on the host)
    if (gid >= (uint32_t)NP * (uint32_t)Dq) return;
    const int p = (int)(gid / (uint32_t)Dq), q = (int)(gid - (uint32_t)p * (uint32_t)Dq);
    const size_t f = blockIdx.y;
    const double* __restrict__ p0 = a.pd0 + f * 2 * (size_t)NP;
    const double* __restrict__ nd = a.nd + f * 2 * (size_t)NP;
    const double bx = __dsub_rn(p0[p], 1.0), by = __dsub_rn(p0[NP + p], 1.0);          // :348-349
    const double ux = nd[p], uy = nd[NP + p];
    const double off = a.off[f * (size_t)NP + p];
    const uint32_t* __restrict__ cen2 = a.cen2 + f * (size_t)NP;
    const uint32_t c1 = a.cen1[f * (size_t)NP + p];
    uint8_t* out = a.Craw + f * (size_t)NP * D + (uint32_t)p * (uint32_t)D + 4u * q;
    const double* __restrict__ vz = a.vz + 4 * q;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (VEC4 || 4 * q + k < D) {
            const double s = __dmul_rn(off, vz[k]);                                        // offset * vzInd
            const double ox = __dmul_rn(s, ux), oy = __dmul_rn(s, uy);                    // :365-366
            int x2 = round_to_i32_x86(__dadd_rn(bx, ox));                                 // :371
            int y2 = round_to_i32_x86(__dadd_rn(by, oy));                                 // :372
            x2 = clampi(x2, 0, W - 1);
            y2 = clampi(y2, 0, H - 1);
            const uint32_t cost = __popc(c1 ^ cen2[y2 * W + x2]);                         // :377-378
            packed |= cost << (8 * k);
        }
    }
    if (VEC4) {
        *(uint32_t*)out = packed;
    } else {
        for (int k = 0; k < 4; k++)
            if (4 * q + k < D) out[k] = (uint8_t)(packed >> (8 * k));
    }
}

// Same for D % 8 == 0, one thread = one pixel, all d.  Neighbouring lanes are neighbouring pixels at the
// same d, so the gather of the second image's census word touches two or three cache lines per wave
// whatever the direction of the epipolar lines (with d across lanes every lane lands in its own line and
// the kernel is bound by the L1 tag rate); vzInd(d) is wave-uniform and comes from scalar loads; the
// per-pixel maps are read once, coalesced.  The 8 costs of a chunk are packed and parked in a per-wave
// LDS tile [64 px][128 B + 8 pad] and written out as full 128-B lines.  ~30 VALU instructions per voxel,
// most of them the reference's own fp64 sequence (3 mul, 2 add, 2 x round-half-away + truncate).
// SMALL: every sample position of the wave is below 2^30 in magnitude (checked per pixel from the maps and
// max |vzInd|), so the rounding needs no range test.
constexpr int RC_SEG = 128, RC_PAD = RC_SEG + 8;
template <bool SMALL>
__device__ __forceinline__ void epi_rawcost_px_body(const EpiCostArgs& a, uint8_t* tilew, const uint32_t pw, const uint32_t p,
                                                    const double bx, const double by, const double ux, const double uy, const double off) {
    const int W = a.W, H = a.H, D = a.D;
    const uint32_t NP = (uint32_t)W * (uint32_t)H;
    const int lane = threadIdx.x & 63;
    const size_t f = blockIdx.y;
    const char* __restrict__ cen2 = (const char*)(a.cen2 + f * (size_t)NP);
    const uint32_t c1 = a.cen1[f * (size_t)NP + p];
    const double* __restrict__ vzt = a.vz;
    const int xhi = W - 1, yhi = H - 1;
    const uint32_t W4 = 4u * (uint32_t)W;
    uint8_t* const myrow = tilew + lane * RC_PAD;
    uint8_t* const outw = a.Craw + f * (size_t)NP * D + (size_t)pw * D;
    // byte offsets of the 8 census words chunk (d0 + c) of this pixel samples: the reference's fp64 sequence per voxel
    auto sample_offsets = [&](const int dbase, uint32_t (&boff)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double s = __dmul_rn(off, vzt[dbase + k]);                                  // offset * vzInd
            const double ox = __dmul_rn(s, ux), oy = __dmul_rn(s, uy);                        // :365-366
            const double vx = __dadd_rn(bx, ox), vy = __dadd_rn(by, oy);
            const int x2 = SMALL ? round_clamp_small(vx, xhi) : clamp0(round_to_i32_x86(vx), xhi);   // :371, :374
            const int y2 = SMALL ? round_clamp_small(vy, yhi) : clamp0(round_to_i32_x86(vy), yhi);   // :372, :375
            boff[k] = __umul24((uint32_t)y2, W4) + ((uint32_t)x2 << 2);                       // 4W, H < 2^24 (launcher)
        }
    };
    for (int d0 = 0; d0 < D; d0 += RC_SEG) {
        const int sb = min(RC_SEG, D - d0);
        // Software pipeline over the chunks of 8 d: the gathered census words of chunk c are requested before the
        // arithmetic of chunk c+1 (some 200 instructions) and consumed after it, so the gather's latency -- L2 hits, a
        // few hundred cycles -- no longer parks the wave once per chunk (round 3 counters: 29 % of the wave time in
        // s_waitcnt with the loads waited for where they were issued).
        uint32_t boff[8], word[8];
        sample_offsets(d0, boff);
#pragma unroll
        for (int k = 0; k < 8; k++) word[k] = *(const uint32_t*)(cen2 + boff[k]);
        for (int c = 0; c < sb; c += 8) {
            uint32_t nword[8];
            if (c + 8 < sb) {                                                                 // wave-uniform
                sample_offsets(d0 + c + 8, boff);
#pragma unroll
                for (int k = 0; k < 8; k++) nword[k] = *(const uint32_t*)(cen2 + boff[k]);
            }
            uint32_t packed[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t cost = __popc(c1 ^ word[k]);                                   // :377-378
                asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(packed[k >> 2]) : "v"(cost), "n"(8 * (k & 3)));
            }
            *(uint2*)(myrow + c) = make_uint2(packed[0], packed[1]);
#pragma unroll
            for (int k = 0; k < 8; k++) word[k] = nword[k];
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e = lane * 8u; e < 64u * (uint32_t)sb; e += 512u) {                     // tile -> HBM, whole lines
            const uint32_t px = e / (uint32_t)sb, o = e - px * (uint32_t)sb;
            if (pw + px < NP) *(uint2*)(outw + (size_t)px * D + d0 + o) = *(const uint2*)(tilew + px * RC_PAD + o);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void epi_rawcost_px_kernel(EpiCostArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[4][64 * RC_PAD];
    const uint32_t NP = (uint32_t)a.W * (uint32_t)a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t pw = (blockIdx.x * 4u + wave) * 64u;          // first pixel of this wave
    if (pw >= NP) return;                                        // wave-uniform
    const uint32_t p = min(pw + lane, NP - 1);                   // lanes past the end redo the last pixel, never stored
    const size_t f = blockIdx.y;
    const double* __restrict__ p0 = a.pd0 + f * 2 * (size_t)NP;
    const double* __restrict__ nd = a.nd + f * 2 * (size_t)NP;
    const double bx = __dsub_rn(p0[p], 1.0), by = __dsub_rn(p0[NP + p], 1.0);          // :348-349
    const double ux = nd[p], uy = nd[NP + p];
    const double off = a.off[f * (size_t)NP + p];
    // |off * vz * u| <= reach * (1 + 3 ulp); NaN or inf anywhere fails the compares
    const double reach = __dmul_rn(__dmul_rn(fabs(off), a.vzmax), fmax(fabs(ux), fabs(uy)));
    const bool small = fabs(bx) < 536870912.0 && fabs(by) < 536870912.0 && reach < 536870912.0 &&
                       fabs(ux) <= 1.7e308 && fabs(uy) <= 1.7e308;
    if (__builtin_amdgcn_ballot_w64(!small) == 0) epi_rawcost_px_body<true>(a, tile[wave], pw, p, bx, by, ux, uy, off);
    else                                          epi_rawcost_px_body<false>(a, tile[wave], pw, p, bx, by, ux, uy, off);
}

// =============================================================================================
// 5x5 box mean, replicate border  (calc_cost_sgm.cpp:387-407).
// (u8)(1.0*sum/25 + 0.5) == (2*sum + 25) / 50 in integers: the exact value has denominator 50,
// so it is never within 1/50 of an integer from below except at the integer itself, which
// needs 2*sum+25 (odd) divisible by 50 -- impossible.
// One thread = one pixel x 4 consecutive d (SWAR: even/odd bytes summed in 16-bit fields).
// =============================================================================================
__global__ __launch_bounds__(256) void box5x5_kernel(const uint8_t* __restrict__ Craw,
                                                     uint8_t* __restrict__ C, int W, int H, int D) {
    const int NP = W * H;
    const int Dq = (D + 3) >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)NP * Dq) return;
    const int p = (int)(gid / Dq), q = (int)(gid - (long long)p * Dq);
    const int y = p / W, x = p - y * W;
    const uint8_t* raw = Craw + (size_t)blockIdx.y * NP * D;
    uint8_t* out = C + (size_t)blockIdx.y * NP * D + (size_t)p * D + 4 * q;
    if ((D & 3) == 0) {
        uint32_t se = 0, so = 0;
#pragma unroll
        for (int dy = -2; dy <= 2; dy++) {
            const int y1 = clampi(y + dy, 0, H - 1);
#pragma unroll
            for (int dx = -2; dx <= 2; dx++) {
                const int x1 = clampi(x + dx, 0, W - 1);
                const uint32_t w = *(const uint32_t*)(raw + ((size_t)y1 * W + x1) * D + 4 * q);
                se += w & 0x00FF00FFu;
                so += (w >> 8) & 0x00FF00FFu;
            }
        }
        const uint32_t b0 = (2 * (se & 0xFFFF) + 25) / 50, b2 = (2 * (se >> 16) + 25) / 50;
        const uint32_t b1 = (2 * (so & 0xFFFF) + 25) / 50, b3 = (2 * (so >> 16) + 25) / 50;
        *(uint32_t*)out = (b0 & 0xFF) | ((b1 & 0xFF) << 8) | ((b2 & 0xFF) << 16) | ((b3 & 0xFF) << 24);
    } else {
        for (int k = 0; k < 4; k++) {
            const int d = 4 * q + k;
            if (d >= D) break;
            uint32_t s = 0;
            for (int dy = -2; dy <= 2; dy++) {
                const int y1 = clampi(y + dy, 0, H - 1);
                for (int dx = -2; dx <= 2; dx++) {
                    const int x1 = clampi(x + dx, 0, W - 1);
                    s += raw[((size_t)y1 * W + x1) * D + d];
                }
            }
            out[k] = (uint8_t)((2 * s + 25) / 50);
        }
    }
}

// =============================================================================================
// 5x5 box mean, sliding-window form for D % 4 == 0 (same result as box5x5_kernel).
// One thread = one pixel column x 4 consecutive d, walking down BOX_ROWS rows: per row one
// horizontal 5-sum (5 u32 loads, even/odd bytes in 16-bit fields) and a 5-deep ring of those sums
// for the vertical part: 5 loads per output instead of 25.
// =============================================================================================
constexpr int BOX_ROWS = 32;
__global__ __launch_bounds__(256) void box5x5_sliding_kernel(const uint8_t* __restrict__ Craw,
                                                             uint8_t* __restrict__ C, int W, int H, int D) {
    const int Dq = D >> 2;
    const int cols = 256 / Dq;                               // pixel columns per block (Dq <= 256 here)
    const int q = threadIdx.x % Dq, xi = threadIdx.x / Dq;
    const int x = blockIdx.x * cols + xi;
    if (xi >= cols || x >= W) return;
    const int y0 = blockIdx.y * BOX_ROWS, y1 = min(y0 + BOX_ROWS, H);
    const size_t NP = (size_t)W * H;
    const uint8_t* raw = Craw + (size_t)blockIdx.z * NP * D + 4 * q;
    uint8_t* out = C + (size_t)blockIdx.z * NP * D + 4 * q;
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xs[k] = clampi(x + k - 2, 0, W - 1);
    auto hsum = [&](int y, uint32_t& he, uint32_t& ho) {     // horizontal 5-sum of row clamp(y)
        const uint8_t* r = raw + (size_t)clampi(y, 0, H - 1) * W * D;
        he = 0; ho = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t w = *(const uint32_t*)(r + (size_t)xs[k] * D);
            he += w & 0x00FF00FFu;
            ho += (w >> 8) & 0x00FF00FFu;
        }
    };
    uint32_t re[5], ro[5];                                   // ring: rows y-2 .. y+2 around the output row
#pragma unroll
    for (int k = 0; k < 4; k++) hsum(y0 - 2 + k, re[k], ro[k]);
    for (int yb = y0; yb < y1; yb += 5) {
#pragma unroll
        for (int k = 0; k < 5; k++) {                        // static ring slot = (k + 4) % 5
            const int y = yb + k;
            if (y < y1) {
                hsum(y + 2, re[(k + 4) % 5], ro[(k + 4) % 5]);
                const uint32_t se = re[0] + re[1] + re[2] + re[3] + re[4];
                const uint32_t so = ro[0] + ro[1] + ro[2] + ro[3] + ro[4];
                const uint32_t b0 = (2 * (se & 0xFFFF) + 25) / 50, b2 = (2 * (se >> 16) + 25) / 50;
                const uint32_t b1 = (2 * (so & 0xFFFF) + 25) / 50, b3 = (2 * (so >> 16) + 25) / 50;
                *(uint32_t*)(out + ((size_t)y * W + x) * D) = (b0 & 0xFF) | ((b1 & 0xFF) << 8) | ((b2 & 0xFF) << 16) | ((b3 & 0xFF) << 24);
            }
        }
    }
}

// =============================================================================================
// The same sliding window with 16 bytes per lane (D % 16 == 0): one thread = one pixel column x 16 consecutive d, D/16
// adjacent lanes a pixel -- every access a 16-byte one, a wave's load 64/(D/16) whole pixels (the dword form above moves
// 4 bytes a lane and reaches 3.5 TB/s; 16-byte accesses are what this chip's memory pipeline is built for).
// =============================================================================================
__global__ __launch_bounds__(256) void box5x5_sliding16_kernel(const uint8_t* __restrict__ Craw,
                                                               uint8_t* __restrict__ C, int W, int H, int D) {
    const int Dq = D >> 4;                                   // lanes per pixel
    const int cols = 256 / Dq;                               // pixel columns per block (Dq <= 64 here)
    const int q = threadIdx.x % Dq, xi = threadIdx.x / Dq;
    const int x = blockIdx.x * cols + xi;
    if (xi >= cols || x >= W) return;
    const int y0 = blockIdx.y * BOX_ROWS, y1 = min(y0 + BOX_ROWS, H);
    const size_t NP = (size_t)W * H;
    const uint8_t* raw = Craw + (size_t)blockIdx.z * NP * D + 16 * q;
    uint8_t* out = C + (size_t)blockIdx.z * NP * D + 16 * q;
    uint32_t xo[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xo[k] = (uint32_t)clampi(x + k - 2, 0, W - 1) * (uint32_t)D;
    struct Sum { uint32_t e[4], o[4]; };                     // even / odd bytes of the 4 dwords in 16-bit fields
    auto hsum = [&](int y, Sum& h) {                         // horizontal 5-sum of row clamp(y)
        const uint8_t* r = raw + (size_t)clampi(y, 0, H - 1) * W * D;
#pragma unroll
        for (int i = 0; i < 4; i++) { h.e[i] = 0; h.o[i] = 0; }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint4 v = *(const uint4*)(r + xo[k]);       // neighbouring lanes re-read these lines: through the L1
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; i++) { h.e[i] += w[i] & 0x00FF00FFu; h.o[i] += (w[i] >> 8) & 0x00FF00FFu; }
        }
    };
    Sum ring[5];                                             // rows y-2 .. y+2 around the output row
#pragma unroll
    for (int k = 0; k < 4; k++) hsum(y0 - 2 + k, ring[k]);
    for (int yb = y0; yb < y1; yb += 5) {
#pragma unroll
        for (int k = 0; k < 5; k++) {                        // static ring slot = (k + 4) % 5
            const int y = yb + k;
            if (y < y1) {
                hsum(y + 2, ring[(k + 4) % 5]);
                uint32_t o4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t se = ring[0].e[i] + ring[1].e[i] + ring[2].e[i] + ring[3].e[i] + ring[4].e[i];
                    const uint32_t so = ring[0].o[i] + ring[1].o[i] + ring[2].o[i] + ring[3].o[i] + ring[4].o[i];
                    const uint32_t b0 = (2 * (se & 0xFFFF) + 25) / 50, b2 = (2 * (se >> 16) + 25) / 50;
                    const uint32_t b1 = (2 * (so & 0xFFFF) + 25) / 50, b3 = (2 * (so >> 16) + 25) / 50;
                    o4[i] = (b0 & 0xFF) | ((b1 & 0xFF) << 8) | ((b2 & 0xFF) << 16) | ((b3 & 0xFF) << 24);
                }
                *(uint4*)(out + ((size_t)y * W + x) * D) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            }
        }
    }
}

// =============================================================================================
// Path aggregation, packed kernel  (calc_cost_sgm.cpp:33-66 sgm_step, :86-257 sgm).
//
// One launch covers every path direction of every frame.  A path direction r has lines
// (rows for the horizontal paths, columns for the vertical ones, wrapped diagonals for the
// oblique ones); every line is an independent serial recurrence.  Lane layout inside a wave:
//   LPP = D/16 adjacent lanes own one pixel, 16 consecutive d each (one 16-byte load);
//   64/LPP adjacent lines advance in lockstep in one wave.
// The 16 costs of a lane live in 8 VGPRs as packed 2 x u16, split into even/odd bytes of each
// loaded dword:  E[k] = (d[4k], d[4k+2]),  O[k] = (d[4k+1], d[4k+3]).  With that split the
// d+1 neighbours of E[k] are O[k] itself and the d-1 neighbours of O[k] are E[k]; the other two
// neighbour vectors cost one v_alignbit each, and only the two lane-boundary elements need a DPP
// row shift.  The running minimum over d is a packed min tree + 3 DPP butterflies (LPP = 8).
//
// Path start (calc_cost_sgm.cpp:152-180): L = C and the stored minimum is 0 (not min C).
// A diagonal line that leaves the image re-enters at the opposite border as a new path, so all
// diagonal lines have H steps and a start wherever x hits the border.
//
// Pass 1 of the reference is the point mirror of pass 0 (:115-123), i.e. pixel index
// NP-1-idx; d is not mirrored.
//
// WRAP=false requires 0<=P1, 0<=P2, max(C)+P2+max(P1,P2) <= 255: then none of the reference's
// u8 narrowings changes a value and 16-bit lanes are exact.  WRAP=true reduces mod 256 at every
// point where the reference narrows to unsigned char.
// =============================================================================================
// the line kernels' path volumes past the caches (stores here, loads in wta_packed_kernel): aggregation 0.236 -> 0.210 ms for one
// 1242x375x128 frame, 0.73 -> 0.645 for four; WTA 0.083 -> 0.076 (three runs each)
#ifndef FSGM_LINE_NT
#define FSGM_LINE_NT 1
#endif
#ifndef FSGM_AGG_PFX
#define FSGM_AGG_PFX 16
#endif
#ifndef FSGM_AGG_PFO
#define FSGM_AGG_PFO 4
#endif
template <int D, int DPL, bool WRAP, int BASE>
__device__ __forceinline__ void agg_packed_body(const AggArgs& a, const int slot, const bool mirror) {
    constexpr int LPP = D / DPL;        // lanes per pixel
    constexpr int NK = DPL / 4;         // cost dwords per lane
    constexpr int PXW = 64 / LPP;       // lines per wave
    // prefetch depth (steps of C in flight per lane): the finely split along-x lines take a dword a step, and a step of theirs is
    // shorter than a quarter of the memory latency
    constexpr int PF = (BASE == 0 && NK == 1) ? FSGM_AGG_PFX : FSGM_AGG_PFO;
    constexpr uint32_t SENT = 0xFFFFFFFFu;
    constexpr uint32_t MASK = 0x00FF00FFu;
    static_assert(LPP >= 1 && LPP <= 32 && (NK == 1 || NK == 2 || NK == 4), "lane split");
    struct __attribute__((aligned(NK * 4))) Words { uint32_t v[NK]; };

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPP, j = lane % LPP;

    const int W = a.W, H = a.H;
    const int NP = W * H;
    const int nlines = BASE == 0 ? H : W;
    const int len = BASE == 0 ? W : H;
    const int lg = ((int)blockIdx.x - a.blk_begin[slot]) * 4 + wave;
    if (lg * PXW >= nlines) return;                         // wave-uniform
    // Lanes past the last line redo the last line: they load the same bytes and store the same
    // bytes to the same addresses as its real lanes (which sit in this same wave), so the loop
    // needs no per-lane predicate.
    const int l = min(lg * PXW + g, nlines - 1);

    const uint8_t* __restrict__ Cf = a.C + (size_t)blockIdx.y * a.c_frame_stride;
    uint8_t* __restrict__ Lf = a.L + (size_t)blockIdx.y * a.l_frame_stride + (size_t)slot * a.l_dir_stride;

    // forward-frame cursors: (x, pix) for the compute side, (xl, pixl) PF steps ahead for loads.
    // pix is the pixel index in the pass-0 frame; pass 1 (mirror) addresses NP-1-pix.
    int x = BASE >= 2 ? l : 0, pix = BASE == 0 ? l * W : l;
    int xl = x, pixl = pix;

    auto advance = [&](int& cx, int& cpix) {               // branch-free
        if (BASE == 0) cpix += 1;
        if (BASE == 1) cpix += W;
        if (BASE == 2) { cx += 1; const bool wr = cx == W; cx = wr ? 0 : cx; cpix += wr ? 1 : W + 1; }
        if (BASE == 3) { cx -= 1; const bool wr = cx < 0; cx = wr ? W - 1 : cx; cpix += wr ? 2 * W - 1 : W - 1; }
    };
    auto byte_off = [&](int cpix) -> uint32_t {
        const int ap = mirror ? NP - 1 - cpix : cpix;
        return (uint32_t)ap * D + (uint32_t)j * DPL;
    };
    // the load cursor runs PF steps ahead of the line's end: keep its address inside the volume
    auto load_c = [&](int cpix) -> Words { return *(const Words*)(Cf + byte_off(min(cpix, NP - 1))); };
    // the path volumes are written once and read once by the WTA kernel: streamed past the caches (FSGM_LINE_NT)
    auto store_words = [&](uint8_t* p, const Words& w) {
        if (FSGM_LINE_NT) {
            typedef uint32_t wv __attribute__((ext_vector_type(NK)));
            wv t;
#pragma unroll
            for (int k = 0; k < NK; k++) t[k] = w.v[k];
            __builtin_nontemporal_store(t, (wv*)p);
        } else *(Words*)p = w;
    };

    const uint32_t P1pk = WRAP ? (uint32_t)(a.P1 & 0xFF) * 0x10001u : (uint32_t)a.P1 * 0x10001u;
    const uint32_t P2pk = (uint32_t)a.P2 * 0x10001u;        // NOWRAP only
    const uint32_t P2b = (uint32_t)(a.P2 & 0xFF);           // WRAP only

    uint32_t LE[NK], LO[NK];                                 // previous pixel's path costs
#pragma unroll
    for (int k = 0; k < NK; k++) LE[k] = LO[k] = 0;
    uint32_t m = 0;                                          // previous pixel's stored minimum

    // one DP step on the DPL costs of this lane; returns the packed output dwords
    auto step = [&](const Words cw, const bool start) -> Words {
        uint32_t CE[NK], CO[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) {
            CE[k] = __builtin_amdgcn_perm(0u, cw.v[k], 0x0C020C00u);   // bytes 0,2 -> 2 x u16
            CO[k] = __builtin_amdgcn_perm(0u, cw.v[k], 0x0C030C01u);   // bytes 1,3 -> 2 x u16
        }
        // lane-boundary neighbours (a pixel of 32 lanes spans two DPP rows: whole-wave shifts there)
        uint32_t prevO3 = LPP > 16 ? dpp_mov<DPP_WAVE_SHR1>(SENT, LO[NK - 1]) : dpp_mov<DPP_ROW_SHR1>(SENT, LO[NK - 1]);
        uint32_t nextE0 = LPP > 16 ? dpp_mov<DPP_WAVE_SHL1>(SENT, LE[0]) : dpp_mov<DPP_ROW_SHL1>(SENT, LE[0]);
        if (j == 0) prevO3 = SENT;                   // d = 0 has no d-1      (:47)
        if (j == LPP - 1) nextE0 = SENT;             // d = D-1 has no d+1    (:48)

        const uint32_t mpk = m | (m << 16);
        uint32_t NE[NK], NO[NK];
        if (!WRAP) {
            const uint32_t p2lane = start ? 0u : P2pk;   // min(.,0)=0 -> L = C at a path start
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const uint32_t nbE = pk_min(align16(LO[k], k ? LO[k - 1] : prevO3), LO[k]);
                const uint32_t nbO = pk_min(LE[k], align16(k < NK - 1 ? LE[k + 1] : nextE0, LE[k]));
                const uint32_t tE = pk_min(LE[k], pk_add(nbE, P1pk));
                const uint32_t tO = pk_min(LO[k], pk_add(nbO, P1pk));
                // C + min(t, m+P2) - m  ==  C + min(t-m, P2)   (no wrap: t >= m)
                NE[k] = pk_add(CE[k], pk_min(pk_sub(tE, mpk), p2lane));
                NO[k] = pk_add(CO[k], pk_min(pk_sub(tO, mpk), p2lane));
            }
        } else {
            const uint32_t jump = (m + P2b) & 0xFFu;                       // :46 u8(LpreMin + P2)
            const uint32_t jpk = jump | (jump << 16);
            // mod-256 adds do not commute with min: narrow each neighbour + P1 first (:47-48),
            // and give the two non-existent neighbours the neutral candidate 255.
            const uint32_t noL = j == 0 ? 0x000000FFu : 0u;
            const uint32_t noR = j == LPP - 1 ? 0x00FF0000u : 0u;
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const uint32_t lE = align16(LO[k], k ? LO[k - 1] : prevO3);            // d-1 of E[k]
                const uint32_t rO = align16(k < NK - 1 ? LE[k + 1] : nextE0, LE[k]);   // d+1 of O[k]
                uint32_t cEl = pk_add(lE, P1pk) & MASK;
                uint32_t cOr = pk_add(rO, P1pk) & MASK;
                if (k == 0) cEl |= noL;
                if (k == NK - 1) cOr |= noR;
                const uint32_t cEr = pk_add(LO[k], P1pk) & MASK;                     // d+1 of E[k] = O[k]
                const uint32_t cOl = pk_add(LE[k], P1pk) & MASK;                     // d-1 of O[k] = E[k]
                const uint32_t tE = pk_min(pk_min(LE[k], jpk), pk_min(cEl, cEr));
                const uint32_t tO = pk_min(pk_min(LO[k], jpk), pk_min(cOl, cOr));
                const uint32_t e = pk_sub(pk_add(CE[k], tE), mpk) & MASK;  // :60 mod 256
                const uint32_t o = pk_sub(pk_add(CO[k], tO), mpk) & MASK;
                NE[k] = start ? CE[k] : e;
                NO[k] = start ? CO[k] : o;
            }
        }
        // minimum over d of the new costs (:61,:65); 0 at a path start (:154,:164)
        uint32_t mk[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) mk[k] = pk_min(NE[k], NO[k]);
#pragma unroll
        for (int w = 1; w < NK; w *= 2)
#pragma unroll
            for (int k = 0; k + w < NK; k += 2 * w) mk[k] = pk_min(mk[k], mk[k + w]);
        const uint32_t mm = mk[0];
        uint32_t mx = min(mm & 0xFFFFu, mm >> 16);
        mx = group_min_u32<LPP>(mx);
        m = start ? 0u : mx;
        Words o;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            LE[k] = NE[k]; LO[k] = NO[k];
            o.v[k] = __builtin_amdgcn_perm(NO[k], NE[k], 0x06020400u);  // bytes E.lo, O.lo, E.hi, O.hi
        }
        return o;
    };
    auto is_start = [&](int t) -> bool { return (t == 0) || (BASE == 2 && x == 0) || (BASE == 3 && x == W - 1); };

    Words ring[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) { ring[i] = load_c(pixl); advance(xl, pixl); }

    // steady state: no branches inside, so the PF loads stay in flight across iterations
    int t0 = 0;
    for (; t0 + PF <= len; t0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const Words cw = ring[i];
            ring[i] = load_c(pixl);
            advance(xl, pixl);
            const Words o = step(cw, is_start(t0 + i));
            store_words(Lf + byte_off(pix), o);
            advance(x, pix);
        }
    }
    // tail (len % PF steps): costs already in the ring
#pragma unroll
    for (int i = 0; i < PF - 1; i++) {
        if (t0 + i < len) {
            const Words o = step(ring[i], is_start(t0 + i));
            store_words(Lf + byte_off(pix), o);
            advance(x, pix);
        }
    }
}

// =============================================================================================
// Along-x lines, D = 32 / 64 / 128, no wrap: the chain that bounds a single call (W steps, 2 H lines a frame -- fewer waves than the
// chip has SIMDs, so a step costs what its instructions cost a lone wave: 7.5-10 cycles each, profiles/r02_ubench_pk_rates.txt).
// Two costs a lane (D = 128: one line a wave; 64: two; 32: four), and the step written out by hand, 18 vector instructions (17 / 21
// at D = 32 / 64) where the general body above takes 55:
//  * the minimum over d comes out of the reduction as a replicated 2 x u16 in a scalar register (v_pk_min with op_sel folds
//    the halves, four row DPP stages, row_bcast:15/31, v_readlane): no broadcast, no select, usable as an operand directly;
//  * the neighbours L[d-1] + P1, L[d+1] + P1 come from whole-wave shifts into two registers whose lane 0 / lane 63 hold the
//    "no such neighbour" value for the whole line (a shift never writes them), and min(L[d], L[d-1] + P1, L[d+1] + P1) is one
//    v_pk_minimum3_f16 whose op_sel picks the halves (exact on these denormal patterns: self-tested at plan creation);
//  * the half of the next step that does not need the minimum fills the wait states between the reduction's DPP stages;
//  * addresses are immediate offsets from one register per 32 steps (32 x 128 bytes = the 4 KB immediate range).
// (D = 64: the minimum of a line's two rows through v_permlane16_swap into a vector register, the cross-line lanes of the shifts
// masked; D = 32: row shifts and the four row stages only.)  Same results as agg_packed_body<D, 4, false, 0>.
// =============================================================================================
#ifndef FSGM_AGG_XLEAN
#define FSGM_AGG_XLEAN 1
#endif

template <int D, bool MIRROR>
__device__ __forceinline__ void agg_x_lean_body(const AggArgs& a, const int slot) {
    constexpr int LPP = D / 2, PXW = 64 / LPP, PF = 32;        // two costs a lane; lines per wave
    static_assert(D == 32 || D == 64 || D == 128, "hand-written along-x step: 16, 32 or 64 lanes a line");
    constexpr int STEP = MIRROR ? -D : D;
    constexpr uint32_t SENT = 0x03FF03FFu;                     // "no neighbour": above every L + P1 (<= 510), below the fp16 normals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPP, j = lane % LPP;
    const int W = a.W, H = a.H;
    const int lg = ((int)blockIdx.x - a.blk_begin[slot]) * 4 + wave;
    if (lg * PXW >= H) return;                                 // wave-uniform
    const int line = min(lg * PXW + g, H - 1);                 // lanes past the last line redo it (same loads, same stores)
    // pass 0 walks row `line` left to right; the mirrored pass walks pixel NP-1-pix: row H-1-line right to left
    const size_t first = MIRROR ? (size_t)W * H - 1 - (size_t)line * W : (size_t)line * W;
    const uint8_t* __restrict__ cp = a.C + (size_t)blockIdx.y * a.c_frame_stride + first * D + 2 * j;
    uint8_t* __restrict__ lp = a.L + (size_t)blockIdx.y * a.l_frame_stride + (size_t)slot * a.l_dir_stride + first * D + 2 * j;

    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2pk = (uint32_t)a.P2 * 0x10001u;
    const uint32_t selc = 0x0C010C00u;     // u16 {c0, c1} -> 2 x u16
    const uint32_t selo = 0x0C0C0200u;     // 2 x u16 -> u16 of two bytes
    const uint32_t mL = j == 0 ? SENT : 0u, mR = j == LPP - 1 ? SENT : 0u;    // (32 lanes a line: the whole-wave shifts cross into the other line)
    uint32_t T = 0, CE, PR = SENT, NR = SENT, sm = 0, M = 0;

    // step t: phase B (needs the previous minimum) + reduction, with phase A of step t+1 in the reduction's wait states.
    // T: min(L[d], L[d-1] + P1, L[d+1] + P1) of the previous pixel; CE: this pixel's costs; cn: the next pixel's, raw.
    // The previous minimum, replicated in both halves: a scalar register with 64 lanes a line (M unused), the vector register M below.
#define FSGM_XLEAN_HEAD(MIN) \
            "v_pk_sub_u16 %[T], %[T], " MIN "\n\t" \
            "v_pk_min_u16 %[T], %[T], %[P2]\n\t" \
            "v_pk_add_u16 %[CUR], %[T], %[CE]\n\t" \
            "v_pk_min_u16 %[X], %[CUR], %[CUR] op_sel:[0,1] op_sel_hi:[1,0]\n\t" \
            "v_pk_add_u16 %[CP], %[CUR], %[P1]\n\t" \
            "v_perm_b32 %[O], %[CUR], %[CUR], %[SELO]\n\t" \
            "v_min_u32_dpp %[X], %[X], %[X] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
            "v_perm_b32 %[CE], %[CN], %[CN], %[SELC]\n\t"
    auto block = [&](const uint32_t cn, const uint32_t p2) -> uint32_t {
        uint32_t CUR, X, O, CP, NB;
        if constexpr (LPP == 64) {
            asm volatile(
                FSGM_XLEAN_HEAD("%[SM]")
                "v_mov_b32_dpp %[PR], %[CP] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %[NR], %[CP] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 0\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_alignbit_b32 %[NB], %[NR], %[PR], 16\n\t"
                "s_nop 0\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_pk_minimum3_f16 %[T], %[CUR], %[CP], %[NB] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
                "s_nop 0\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_readlane_b32 %[SM], %[X], 63\n\t"
                "s_nop 1"
                : [T] "+v"(T), [CE] "+v"(CE), [PR] "+v"(PR), [NR] "+v"(NR), [SM] "+s"(sm),
                  [CUR] "=&v"(CUR), [X] "=&v"(X), [O] "=&v"(O), [CP] "=&v"(CP), [NB] "=&v"(NB)
                : [CN] "v"(cn), [P2] "s"(p2), [P1] "s"(P1pk), [SELO] "s"(selo), [SELC] "s"(selc));
        } else if constexpr (LPP == 32) {
            // two lines a wave: the shifted-in neighbours of a line's first / last lane are masked to "none", the minimum of a
            // line's two rows meets through a row swap and stays in a vector register
            uint32_t Y;
            asm volatile(
                FSGM_XLEAN_HEAD("%[M]")
                "v_mov_b32_dpp %[PR], %[CP] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %[NR], %[CP] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_or_b32 %[PR], %[PR], %[ML]\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_or_b32 %[NR], %[NR], %[MR]\n\t"
                "v_alignbit_b32 %[NB], %[NR], %[PR], 16\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_pk_minimum3_f16 %[T], %[CUR], %[CP], %[NB] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
                "s_nop 0\n\t"
                "v_mov_b32 %[Y], %[X]\n\t"
                "s_nop 1\n\t"
                "v_permlane16_swap_b32 %[X], %[Y]\n\t"
                "s_nop 0\n\t"
                "v_min_u32 %[M], %[X], %[Y]"
                : [T] "+v"(T), [CE] "+v"(CE), [PR] "+v"(PR), [NR] "+v"(NR), [M] "+v"(M),
                  [CUR] "=&v"(CUR), [X] "=&v"(X), [Y] "=&v"(Y), [O] "=&v"(O), [CP] "=&v"(CP), [NB] "=&v"(NB)
                : [CN] "v"(cn), [P2] "s"(p2), [P1] "s"(P1pk), [SELO] "s"(selo), [SELC] "s"(selc), [ML] "v"(mL), [MR] "v"(mR));
        } else {
            // four lines a wave, a row each: row shifts never write a row's first / last lane, the row stages are the whole reduction
            asm volatile(
                FSGM_XLEAN_HEAD("%[M]")
                "v_mov_b32_dpp %[PR], %[CP] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %[NR], %[CP] row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 0\n\t"
                "v_min_u32_dpp %[X], %[X], %[X] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_alignbit_b32 %[NB], %[NR], %[PR], 16\n\t"
                "s_nop 0\n\t"
                "v_min_u32_dpp %[M], %[X], %[X] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_pk_minimum3_f16 %[T], %[CUR], %[CP], %[NB] op_sel:[0,1,0] op_sel_hi:[1,0,1]"
                : [T] "+v"(T), [CE] "+v"(CE), [PR] "+v"(PR), [NR] "+v"(NR), [M] "+v"(M),
                  [CUR] "=&v"(CUR), [X] "=&v"(X), [O] "=&v"(O), [CP] "=&v"(CP), [NB] "=&v"(NB)
                : [CN] "v"(cn), [P2] "s"(p2), [P1] "s"(P1pk), [SELO] "s"(selo), [SELC] "s"(selc));
        }
        return O;
    };
#undef FSGM_XLEAN_HEAD
    auto ld = [&](const int t) -> uint32_t { return *(const uint16_t*)(cp + (ptrdiff_t)t * STEP); };
    auto st = [&](const int t, const uint32_t o) {
        if (FSGM_LINE_NT) __builtin_nontemporal_store((uint16_t)o, (uint16_t*)(lp + (ptrdiff_t)t * STEP));
        else *(uint16_t*)(lp + (ptrdiff_t)t * STEP) = (uint16_t)o;
    };

    // path start (calc_cost_sgm.cpp:154,164): L = C, stored minimum 0
    CE = __builtin_amdgcn_perm(0u, ld(0), selc);
    uint32_t ring[PF];                                         // ring[i]: the costs of step u0 + i + 2 (u = t - 1)
#pragma unroll
    for (int i = 0; i < PF; i++) ring[i] = ld(min(i + 2, W - 1));
    st(0, block(ld(min(1, W - 1)), 0u));
    sm = 0; M = 0;

    const int n = W - 1;                                       // steps u = 0..n-1 are pixels t = u + 1
    int u0 = 0;
    for (; u0 + 2 * PF + 2 <= W; u0 += PF) {                   // every load of the chunk inside the line; at most 2 PF steps are left after it
        const uint8_t* cq = cp + (ptrdiff_t)(u0 + 2 + PF) * STEP;
        uint8_t* lq = lp + (ptrdiff_t)(u0 + 1) * STEP;
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const uint32_t o = block(ring[i], P2pk);
            ring[i] = *(const uint16_t*)(cq + i * STEP);
            if (FSGM_LINE_NT) __builtin_nontemporal_store((uint16_t)o, (uint16_t*)(lq + i * STEP));
            else *(uint16_t*)(lq + i * STEP) = (uint16_t)o;
        }
    }
#pragma unroll
    for (int i = 0; i < PF; i++) {
        if (u0 + i < n) {
            const uint32_t cn = ring[i];
            if (u0 + i + 2 + PF < W) ring[i] = ld(u0 + i + 2 + PF);
            st(u0 + i + 1, block(cn, P2pk));
        }
    }
    u0 += PF;
#pragma unroll
    for (int i = 0; i < PF; i++)
        if (u0 + i < n) st(u0 + i + 1, block(ring[i], P2pk));
}

// =============================================================================================
// The other three directions (along y and the two diagonals), no wrap, 16 costs a lane: the general body with what a
// step does not need taken out -- these lines are most of a single call's instructions (6 of 8 slots), and with a few
// waves a SIMD an instruction costs 5-8 cycles whatever it does:
//  * min(L[d], L[d-1] + P1, L[d+1] + P1) as one v_pk_minimum3_f16 a register (exact on these denormal patterns: self-tested
//    at plan creation) on L + P1 aligned once, instead of align / min / add / min;
//  * the minimum over d leaves the reduction replicated in both halves (v_pk_min with op_sel folds them): no extract, no
//    re-broadcast;
//  * byte offsets advance by a constant (along y) or by one of two constants chosen by a down-counter (diagonals), the loads
//    stop at the line's end instead of being clamped, and a path start is the wrap of the previous advance.
// Same results as agg_packed_body<D, 16, false, BASE> (tests/test_gpu_epi.py runs both).
// =============================================================================================
#ifndef FSGM_AGG_LEAN
#define FSGM_AGG_LEAN 1
#endif
__device__ __forceinline__ uint32_t pk_min_fold(uint32_t x) {       // min(x.lo, x.hi) in both halves
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
    return r;
}
template <int D, int BASE>
__device__ __forceinline__ void agg_lean_body(const AggArgs& a, const int slot, const bool mirror) {
    constexpr int LPP = D / 16, PXW = 64 / LPP, PF = 4;
    constexpr uint32_t SENT = 0x03FF03FFu, MASK = 0x00FF00FFu;   // "no neighbour": above every L + P1, a denormal pattern
    static_assert(BASE >= 1 && BASE <= 3 && LPP >= 2 && LPP <= 16, "lean body: along y / diagonals, D = 32..256");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPP, j = lane % LPP;
    const int W = a.W, H = a.H, NP = W * H, len = H;
    const int lg = ((int)blockIdx.x - a.blk_begin[slot]) * 4 + wave;
    if (lg * PXW >= W) return;                               // wave-uniform
    const int l = min(lg * PXW + g, W - 1);                  // lanes past the last line redo it (same loads, same stores)

    const uint8_t* __restrict__ Cf = a.C + (size_t)blockIdx.y * a.c_frame_stride;
    uint8_t* __restrict__ Lf = a.L + (size_t)blockIdx.y * a.l_frame_stride + (size_t)slot * a.l_dir_stride;
    // byte offset of this lane's 16 costs at the pass-0 pixel `pix`; the mirrored pass addresses NP-1-pix
    const int sgn = mirror ? -1 : 1;
    const uint32_t dN = (uint32_t)(sgn * (W + (BASE == 2 ? 1 : BASE == 3 ? -1 : 0)) * D);    // a step inside the frame
    const uint32_t dW = (uint32_t)(sgn * (BASE == 2 ? 1 : 2 * W - 1) * D);                   // the step that wraps in x
    uint32_t oc = (uint32_t)(mirror ? NP - 1 - l : l) * D + (uint32_t)j * 16, ol = oc;        // compute / load cursors
    int cc = BASE == 2 ? W - l : l + 1, cl = cc;                                             // advances until the wrap
    // advance a cursor; true where it wrapped (the pixel reached starts a new path, calc_cost_sgm.cpp:154-177)
    auto adv = [&](uint32_t& o, int& c) -> bool {
        if (BASE == 1) { o += dN; return false; }
        c -= 1;
        const bool wr = c == 0;
        c = wr ? W : c;
        o += wr ? dW : dN;
        return wr;
    };
    auto load_c = [&](uint32_t o) -> uint4 { return *(const uint4*)(Cf + o); };
    auto store_l = [&](uint32_t o, const uint4& v) {
        if (FSGM_LINE_NT) store_nt(Lf + o, v); else *(uint4*)(Lf + o) = v;
    };

    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2pk = (uint32_t)a.P2 * 0x10001u;
    const uint32_t mL = j == 0 ? SENT : 0u, mR = j == LPP - 1 ? SENT : 0u;
    uint32_t LE[4], LO[4], m = 0, pr = SENT, nr = SENT;
#pragma unroll
    for (int k = 0; k < 4; k++) LE[k] = LO[k] = 0;

    // one step; FIRST: the line's first pixel (every lane starts a path); start: lanes whose previous advance wrapped
    auto step = [&](const uint4 raw, auto first, const bool start) -> uint4 {
        constexpr bool FIRST = decltype(first)::value;
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
        uint32_t CE[4], CO[4], NE[4], NO[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { CE[k] = w[k] & MASK; CO[k] = __builtin_amdgcn_perm(0u, w[k], 0x0C030C01u); }
        if (FIRST) {
#pragma unroll
            for (int k = 0; k < 4; k++) { NE[k] = CE[k]; NO[k] = CO[k]; }
            m = 0;
        } else {
            uint32_t EP[4], OP[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { EP[k] = pk_add(LE[k], P1pk); OP[k] = pk_add(LO[k], P1pk); }
            // the neighbouring lanes' d-1 / d+1 (the first lane of a row of 16 keeps SENT: a row shift never writes it)
            asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pr) : "v"(OP[3]));
            asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(nr) : "v"(EP[0]));
            const uint32_t prevOP = LPP == 16 ? pr : (pr | mL), nextEP = LPP == 16 ? nr : (nr | mR);
            const uint32_t p2lane = (BASE != 1 && start) ? 0u : P2pk;       // min(., 0) = 0: L = C at a path start
            uint32_t mk[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t tE = pk_min3(LE[k], align16(OP[k], k ? OP[k - 1] : prevOP), OP[k]);
                const uint32_t tO = pk_min3(LO[k], EP[k], align16(k < 3 ? EP[k + 1] : nextEP, EP[k]));
                NE[k] = pk_add(CE[k], pk_min(pk_sub(tE, m), p2lane));      // C + min(t - m, P2): t >= m without wrap
                NO[k] = pk_add(CO[k], pk_min(pk_sub(tO, m), p2lane));
                mk[k] = pk_min(NE[k], NO[k]);
            }
            const uint32_t mx = group_min_u32<LPP>(pk_min_fold(pk_min(pk_min(mk[0], mk[1]), pk_min(mk[2], mk[3]))));
            m = (BASE != 1 && start) ? 0u : mx;                            // stored minimum 0 at a path start
        }
        uint4 o;
        uint32_t ow[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            LE[k] = NE[k]; LO[k] = NO[k];
            ow[k] = __builtin_amdgcn_perm(NO[k], NE[k], 0x06020400u);       // bytes E.lo, O.lo, E.hi, O.hi
        }
        o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
        return o;
    };

    // first pixel, then steps u = t - 1 = 0..n-1 with the ring PF steps ahead
    store_l(oc, step(load_c(ol), std::true_type{}, true));
    bool wrapped = adv(oc, cc);
    adv(ol, cl);
    const int n = len - 1;
    uint4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) {
        if (i < n) { ring[i] = load_c(ol); adv(ol, cl); } else ring[i] = make_uint4(0, 0, 0, 0);
    }
    int u0 = 0;
    for (; u0 + 2 * PF <= n; u0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const uint4 cw = ring[i];
            ring[i] = load_c(ol);
            adv(ol, cl);
            store_l(oc, step(cw, std::false_type{}, wrapped));
            wrapped = adv(oc, cc);
        }
    }
#pragma unroll
    for (int i = 0; i < PF; i++) {
        if (u0 + i < n) {
            const uint4 cw = ring[i];
            if (u0 + i + PF < n) { ring[i] = load_c(ol); adv(ol, cl); }
            store_l(oc, step(cw, std::false_type{}, wrapped));
            wrapped = adv(oc, cc);
        }
    }
    u0 += PF;
#pragma unroll
    for (int i = 0; i < PF; i++) {
        if (u0 + i < n) {
            store_l(oc, step(ring[i], std::false_type{}, wrapped));
            wrapped = adv(oc, cc);
        }
    }
}

// The along-x lines are the long serial chains (W steps against H for every other direction) and there are
// few of them (2 H lines a frame): split finer over the lanes, 4 costs a lane (8 at D = 256), they take a
// third of the instructions per step; every other direction keeps 16 costs a lane.
#ifndef FSGM_AGG_DX
#define FSGM_AGG_DX 4
#endif
#ifndef FSGM_AGG_DO
#define FSGM_AGG_DO 16
#endif
template <int D> struct AggSplit {
    static constexpr int along_x = D >= 256 && FSGM_AGG_DX < 8 ? 8 : FSGM_AGG_DX;
    static constexpr int other = D >= 256 && FSGM_AGG_DO < 8 ? 8 : FSGM_AGG_DO;
};

template <int D, bool WRAP, bool FINE>
__global__ __launch_bounds__(256) void agg_packed_kernel(AggArgs a) {
    // which direction slot does this block belong to (block-uniform)
    int slot = 0;
#pragma unroll
    for (int i = 1; i < 8; i++)
        if (i < a.ndirs && (int)blockIdx.x >= a.blk_begin[i]) slot = i;
    const int code = a.dir_code[slot];
    const bool mirror = (code & 4) != 0;
    constexpr int DO = AggSplit<D>::other, DX = FINE ? AggSplit<D>::along_x : DO;
    switch (code & 3) {                 // 0: along x, 1: along y, 2: x+1,y+1, 3: x-1,y+1
        case 0:
            if constexpr ((D == 32 || D == 64 || D == 128) && !WRAP && FINE && FSGM_AGG_XLEAN) {
                if (mirror) agg_x_lean_body<D, true>(a, slot); else agg_x_lean_body<D, false>(a, slot);
            } else agg_packed_body<D, DX, WRAP, 0>(a, slot, mirror);
            break;
#define FSGM_AGG_OTHER(BASE) \
            if constexpr (!WRAP && DO == 16 && D >= 32 && D <= 256 && FSGM_AGG_LEAN) agg_lean_body<D, BASE>(a, slot, mirror); \
            else agg_packed_body<D, DO, WRAP, BASE>(a, slot, mirror);
        case 1: FSGM_AGG_OTHER(1) break;
        case 2: FSGM_AGG_OTHER(2) break;
        default: FSGM_AGG_OTHER(3) break;
#undef FSGM_AGG_OTHER
    }
}

// =============================================================================================
// Path aggregation, generic kernel: any D (<= FSGM_GENERIC_MAX_D), exact u8 semantics.
// One wave per line, Lpre/Lcur in LDS, lanes stride over d.  Correctness path for disparity
// ranges the packed kernel does not cover; not tuned.
// =============================================================================================
__global__ __launch_bounds__(256) void agg_generic_kernel(AggArgs a) {
    __shared__ uint8_t sL[4][2][FSGM_GENERIC_MAX_D + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int slot = 0;
#pragma unroll
    for (int i = 1; i < 8; i++)
        if (i < a.ndirs && (int)blockIdx.x >= a.blk_begin[i]) slot = i;
    const int code = a.dir_code[slot];
    const int base = code & 3;
    const bool mirror = (code & 4) != 0;
    const int W = a.W, H = a.H, D = a.D;
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    const int line = ((int)blockIdx.x - a.blk_begin[slot]) * 4 + wave;
    if (line >= nlines) return;                              // wave-uniform
    const uint8_t* __restrict__ Cf = a.C + (size_t)blockIdx.y * a.c_frame_stride;
    uint8_t* __restrict__ Lf = a.L + (size_t)blockIdx.y * a.l_frame_stride + (size_t)slot * a.l_dir_stride;
    int x = base >= 2 ? line : 0, pix = base == 0 ? line * W : line;
    const int dpix = base == 0 ? 1 : (base == 1 ? W : (base == 2 ? W + 1 : W - 1));
    uint8_t* pre = sL[wave][0];
    uint8_t* cur = sL[wave][1];
    uint32_t m = 0;
    for (int t = 0; t < len; t++) {
        const bool start = (t == 0) || (base == 2 && x == 0) || (base == 3 && x == W - 1);
        const size_t off = (size_t)(mirror ? NP - 1 - pix : pix) * D;
        uint32_t lo = 255;
        const uint32_t jump = (m + (uint32_t)a.P2) & 0xFF;
        for (int d = lane; d < D; d += 64) {
            const uint32_t c = Cf[off + d];
            uint32_t v;
            if (start) {
                v = c;
            } else {
                uint32_t best = min(jump, (uint32_t)pre[d]);
                if (d > 0) best = min(best, ((uint32_t)pre[d - 1] + (uint32_t)a.P1) & 0xFF);
                if (d < D - 1) best = min(best, ((uint32_t)pre[d + 1] + (uint32_t)a.P1) & 0xFF);
                v = (c + best - m) & 0xFF;
            }
            cur[d] = (uint8_t)v;
            Lf[off + d] = (uint8_t)v;
            lo = min(lo, v);
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, s));
        m = start ? 0u : lo;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): LDS writes visible to the wave
        uint8_t* tmp = pre; pre = cur; cur = tmp;
        pix += dpix;
        if (base == 2) { x++; if (x == W) { x = 0; pix -= W; } }
        else if (base == 3) { x--; if (x < 0) { x = W - 1; pix += W; } }
    }
}

// =============================================================================================
// S = sum over paths (debug tap of the reference's Sp, calc_cost_sgm.cpp:227-232)
// =============================================================================================
__global__ __launch_bounds__(256) void sum_paths_kernel(const uint8_t* __restrict__ L, uint32_t* __restrict__ S,
                                                        size_t n, size_t dir_stride, int ndirs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t s = 0;
    for (int r = 0; r < ndirs; r++) s += L[r * dir_stride + i];
    S[i] = s;
}

// =============================================================================================
// WTA + sub-pixel + vz->disparity  (calc_cost_sgm.cpp:259-308, :414-426), shared tail.
// c_1, c, c1 are S[best-1], S[best], S[best+1] (u32 in the reference).
// =============================================================================================
__device__ __forceinline__ uint32_t sum_at(const uint8_t* Lf, size_t dir_stride, int ndirs, size_t idx) {
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) s += r < ndirs ? (uint32_t)Lf[r * dir_stride + idx] : 0u;   // loads issued together
    return s;
}

// packed WTA: LPP lanes per pixel, 16 d per lane; sums kept as 2 x u16 (8 paths x 255 < 65536)
template <int LPP>
__global__ __launch_bounds__(256) void wta_packed_kernel(WtaArgs a) {
    constexpr int D = LPP * 16;
    constexpr int PPB = 256 / LPP;                           // pixels per block
    constexpr uint32_t MASK = 0x00FF00FFu;
    __shared__ uint32_t sS[256 * 8];                         // u16 S[PPB][D]
    const int tid = threadIdx.x;
    const int NP = a.W * a.H;
    const int gp = blockIdx.x * PPB + tid / LPP, j = tid % LPP;
    const bool valid = gp < NP;
    const int p = valid ? gp : NP - 1;
    const size_t f = blockIdx.y;
    const uint8_t* __restrict__ Lf = a.L + f * a.l_frame_stride;
    uint32_t E[4] = {0, 0, 0, 0}, O[4] = {0, 0, 0, 0};
    const size_t off = (size_t)p * D + (size_t)j * 16;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (r < a.ndirs) {
            const uint4 v = FSGM_LINE_NT ? load_nt(Lf + (size_t)r * a.l_dir_stride + off) : *(const uint4*)(Lf + (size_t)r * a.l_dir_stride + off);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) { E[k] += w[k] & MASK; O[k] += (w[k] >> 8) & MASK; }
        }
    }
    // S to LDS in natural d order; key = (S << 8) | d, minimum key = first minimum (:267)
    uint32_t key = 0xFFFFFFFFu;
    uint32_t* row = sS + (size_t)(tid / LPP) * (D / 2) + j * 8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t s0 = E[k] & 0xFFFF, s1 = O[k] & 0xFFFF, s2 = E[k] >> 16, s3 = O[k] >> 16;
        row[2 * k] = s0 | (s1 << 16);
        row[2 * k + 1] = s2 | (s3 << 16);
        const uint32_t d0 = (uint32_t)j * 16 + 4 * k;
        key = min(key, min(min((s0 << 8) | d0, (s1 << 8) | (d0 + 1)), min((s2 << 8) | (d0 + 2), (s3 << 8) | (d0 + 3))));
    }
    key = group_min_u32<LPP>(key);
    __syncthreads();
    if (j == 0 && valid) {
        const uint32_t best = key & 0xFF, minc = key >> 8;
        const uint16_t* srow = (const uint16_t*)(sS + (size_t)(tid / LPP) * (D / 2));
        uint32_t c_1 = 0, c1 = 0;
        if (a.subpixel && best > 1) {
            c_1 = srow[best - 1];
            if (best + 1 < (uint32_t)D) c1 = srow[best + 1];
            else if (p + 1 < NP) c1 = sum_at(Lf, a.l_dir_stride, a.ndirs, (size_t)(p + 1) * D);   // next pixel's d=0 (:296)
            else c1 = 0;                                      // past the end of Sp: defined as 0 here
        }
        wta_finish(a, f, p, best, minc, c_1, c1);
    }
}

// generic WTA: one wave per pixel, lanes stride over d
__global__ __launch_bounds__(256) void wta_generic_kernel(WtaArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NP = a.W * a.H, D = a.D;
    const int p = blockIdx.x * 4 + wave;
    if (p >= NP) return;
    const size_t f = blockIdx.y;
    const uint8_t* __restrict__ Lf = a.L + f * a.l_frame_stride;
    uint32_t key = 0xFFFFFFFFu;
    for (int d = lane; d < D; d += 64) {
        const uint32_t s = sum_at(Lf, a.l_dir_stride, a.ndirs, (size_t)p * D + d);
        key = min(key, (s << 12) | (uint32_t)d);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, s));
    if (lane == 0) {
        const uint32_t best = key & 0xFFF, minc = key >> 12;
        uint32_t c_1 = 0, c1 = 0;
        if (a.subpixel && best > 1) {
            c_1 = sum_at(Lf, a.l_dir_stride, a.ndirs, (size_t)p * D + best - 1);
            if ((size_t)p * D + best + 1 < (size_t)NP * D) c1 = sum_at(Lf, a.l_dir_stride, a.ndirs, (size_t)p * D + best + 1);
        }
        wta_finish(a, f, p, best, minc, c_1, c1);
    }
}

// =============================================================================================
// forward-backward consistency check  (calc_cost_sgm.cpp:429-480 calc_disp_from_first, :482-536
// forward_backward_check; dead code in the shipped reference -- its call at :589-590 is commented
// out -- offered here as an option).  Runs on bestD (vz index * 256) before vz->disparity.
// The reference's scatter "D2[t] = D1 if D2[t] is invalid or smaller" in raster order is a
// maximum over the contributing pixels, so it is done with atomicMax on D1+1 (0 = invalid).
// =============================================================================================
__device__ __forceinline__ double fb_displacement(const FbArgs& a, size_t f, int p, uint32_t d1) {
    const double d = __ddiv_rn((double)d1, 256.0);                                 // :447 / :502
    const double r = __dmul_rn(__ddiv_rn(d, (double)a.n), a.vMax);
    const double vz = __ddiv_rn(r, __dsub_rn(1.0, r));
    return __dmul_rn(a.off[f * (size_t)a.W * a.H + p], vz);                        // :451 / :506
}

__global__ __launch_bounds__(256) void fb_scatter_kernel(FbArgs a) {
    const int NP = a.W * a.H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const size_t f = blockIdx.y;
    const uint32_t d1 = a.D1[f * (size_t)NP + p];
    const double d = fb_displacement(a, f, p, d1);
    const double* p0 = a.pd0 + f * 2 * (size_t)NP;
    const double* nd = a.nd + f * 2 * (size_t)NP;
    const int p2x = f64_to_i32_x86(__dadd_rn(__dsub_rn(p0[p], 1.0), __dmul_rn(d, nd[p])));            // :462
    const int p2y = f64_to_i32_x86(__dadd_rn(__dsub_rn(p0[NP + p], 1.0), __dmul_rn(d, nd[NP + p])));  // :463
    uint32_t* enc = a.D2enc + f * (size_t)NP;
#pragma unroll
    for (int dy = 0; dy <= 1; dy++)
#pragma unroll
        for (int dx = 0; dx <= 1; dx++) {
            const long long tx = (long long)dx + p2x, ty = (long long)dy + p2y;
            if (tx >= 0 && tx < a.W && ty >= 0 && ty < a.H) atomicMax(&enc[ty * a.W + tx], d1 + 1u);    // :471-474
        }
}

__global__ __launch_bounds__(256) void fb_check_kernel(FbArgs a) {
    const int NP = a.W * a.H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const size_t f = blockIdx.y;
    const uint32_t INVALID = 512u << 8;                                            // :5
    const uint32_t* enc = a.D2enc + f * (size_t)NP;
    const uint32_t mine = enc[p];
    a.D2[f * (size_t)NP + p] = mine ? mine - 1u : INVALID;                         // plhs[3] in natural form
    const uint32_t d1 = a.D1[f * (size_t)NP + p];
    const double d = fb_displacement(a, f, p, d1);
    const double* p0 = a.pd0 + f * 2 * (size_t)NP;
    const double* nd = a.nd + f * 2 * (size_t)NP;
    const int p2x = round_to_i32_x86(__dadd_rn(__dsub_rn(p0[p], 1.0), __dmul_rn(d, nd[p])));            // :516
    const int p2y = round_to_i32_x86(__dadd_rn(__dsub_rn(p0[NP + p], 1.0), __dmul_rn(d, nd[NP + p])));  // :517
    uint8_t ok = 1;                                                                // :485
    if (p2x < 0 || p2x > a.W - 1 || p2y < 0 || p2y > a.H - 1) ok = 0;              // :519-522
    else {
        const uint32_t t = enc[p2y * a.W + p2x];
        if (t == 0) ok = 0;                                                        // :524-527
        else {
            long long diff = (long long)(int32_t)d1 - (long long)(int32_t)(t - 1u);   // :529
            if (diff < 0) diff = -diff;
            if (diff > a.thr) ok = 0;
        }
    }
    a.conf[f * (size_t)NP + p] = ok;
}

// convert_vzInd_to_disp as its own pass (calc_cost_sgm.cpp:414-426), used when the check above
// has to see bestD before the conversion
__global__ __launch_bounds__(256) void vz_convert_kernel(uint32_t* __restrict__ bestD, const double* __restrict__ off,
                                                         int NP, int n, double vMax) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const size_t i = (size_t)blockIdx.y * NP + p;
    const double d = __ddiv_rn((double)bestD[i], 256.0);
    const double r = __dmul_rn(__ddiv_rn(d, (double)n), vMax);
    const double vz = __ddiv_rn(r, __dsub_rn(1.0, r));
    bestD[i] = f64_to_u32_x86(__dmul_rn(__dmul_rn(off[i], vz), 256.0));
}

void launch_fb_check(hipStream_t st, const FbArgs& a, int frames) {
    const int NP = a.W * a.H;
    (void)hipMemsetAsync(a.D2enc, 0, (size_t)frames * NP * 4, st);                 // :440-442 all invalid
    dim3 grid((NP + 255) / 256, frames);
    hipLaunchKernelGGL(fb_scatter_kernel, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(fb_check_kernel, grid, dim3(256), 0, st, a);
}

void launch_vz_convert(hipStream_t st, uint32_t* bestD, const double* off, int W, int H, int D, double vMax, int frames) {
    dim3 grid((W * H + 255) / 256, frames);
    hipLaunchKernelGGL(vz_convert_kernel, grid, dim3(256), 0, st, bestD, off, W * H, D + 1, vMax);
}

// =============================================================================================
// launchers
// =============================================================================================
void launch_census(hipStream_t st, const uint8_t* img, uint32_t* cen, int W, int H, int frames) {
    const char* qe = getenv("FSGM_CENSUS_QUAD");                // A/B switch and test hook: 0 never, 1 whenever the shape allows, unset: by size
    const bool quad = !(qe && qe[0] == '0'), force = qe && qe[0] == '1';
    // (a single 1242x375 frame is a latency-bound launch: 466 k threads of the one-pixel kernel finish before 116 k threads of
    // this one -- cost stage 0.060 against 0.064 ms; from a few frames on the instruction count decides: 0.0427 -> 0.0408 at 512)
    if (quad && W >= 16 && H >= 5 && (force || (long long)W * H * frames >= 2000000)) {
        dim3 grid((((W + 3) / 4) * H + 255) / 256, frames);
        hipLaunchKernelGGL(census5x5_quad_kernel, grid, dim3(256), 0, st, img, cen, W, H);
        return;
    }
    dim3 grid((W * H + 255) / 256, frames);
    hipLaunchKernelGGL(census5x5_kernel, grid, dim3(256), 0, st, img, cen, W, H);
}

void launch_epi_cost(hipStream_t st, const EpiCostArgs& a, uint8_t* C, int frames) {
    // FSGM_COST_FUSED=0: the two-kernel form (raw volume through HBM) -- A/B switch and the cross-check of the fused kernel in the tests
    const bool fused = [] { const char* e = getenv("FSGM_COST_FUSED"); return !(e && e[0] == '0'); }();   // read per launch: the tests flip it
    if (fused && costbox_ok(a.W, a.H, a.D)) { launch_epi_costbox(st, a, C, frames); return; }
    const long long n = (long long)a.W * a.H * ((a.D + 3) / 4);
    dim3 grid((unsigned)((n + 255) / 256), frames);
    if ((a.D & 7) == 0 && a.W < (1 << 22) && a.H < (1 << 24)) {
        hipLaunchKernelGGL(epi_rawcost_px_kernel, dim3((unsigned)(((long long)a.W * a.H + 255) / 256), frames), dim3(256), 0, st, a);
    } else if ((a.D & 3) == 0) hipLaunchKernelGGL(epi_rawcost_kernel<true>, grid, dim3(256), 0, st, a);
    else                       hipLaunchKernelGGL(epi_rawcost_kernel<false>, grid, dim3(256), 0, st, a);
    static const bool box16 = [] { const char* e = getenv("FSGM_BOX16"); return !(e && e[0] == '0'); }();   // A/B switch
    if (box16 && (a.D & 15) == 0 && a.D <= 1024) {
        const int cols = 256 / (a.D >> 4);
        dim3 g2((a.W + cols - 1) / cols, (a.H + BOX_ROWS - 1) / BOX_ROWS, frames);
        hipLaunchKernelGGL(box5x5_sliding16_kernel, g2, dim3(256), 0, st, (const uint8_t*)a.Craw, C, a.W, a.H, a.D);
    } else if ((a.D & 3) == 0 && a.D <= 1024) {
        const int cols = 256 / (a.D >> 2);
        dim3 g2((a.W + cols - 1) / cols, (a.H + BOX_ROWS - 1) / BOX_ROWS, frames);
        hipLaunchKernelGGL(box5x5_sliding_kernel, g2, dim3(256), 0, st, (const uint8_t*)a.Craw, C, a.W, a.H, a.D);
    } else {
        hipLaunchKernelGGL(box5x5_kernel, grid, dim3(256), 0, st, (const uint8_t*)a.Craw, C, a.W, a.H, a.D);
    }
}

// Bandwidth probe: one 16-byte element per thread, streaming (non-temporal) loads and stores, as many 256-thread
// workgroups as elements / 256 -- the shape that copies fastest on this chip (tools/ubench/copy_rates.hip: 6.5 TB/s,
// against 5.2-5.5 TB/s for grid-stride loops with 2-8 elements in flight per thread and 4.7-5.4 for hipMemcpyAsync).
__global__ __launch_bounds__(256) void copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) store_nt(dst + i, load_nt(src + i));
}

void launch_copy16(hipStream_t st, void* dst, const void* src, size_t bytes) {
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, n16);
}

int agg_packed_lpp(int D) {
    switch (D) { case 16: return 1; case 32: return 2; case 64: return 4; case 128: return 8; case 256: return 16; }
    return 0;
}

// Fill blk_begin / dir_code for `paths` directions.  Long (horizontal) lines first so that the
// W-step chains start before the H-step ones.
static void plan_dirs(AggArgs& a, int paths, int lines_per_block_x, int lines_per_block) {
    static const int order8[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    static const int order4[4] = {0, 4, 1, 5};      // also the 2-slot horizontal-only plan: {0, 4}
    a.ndirs = paths;
    int acc = 0;
    for (int i = 0; i < paths; i++) {
        const int code = paths == 8 ? order8[i] : order4[i];
        a.dir_code[i] = code;
        a.blk_begin[i] = acc;
        const int nlines = (code & 3) == 0 ? a.H : a.W;
        const int lpb = (code & 3) == 0 ? lines_per_block_x : lines_per_block;
        acc += (nlines + lpb - 1) / lpb;
    }
    for (int i = paths; i <= 8; i++) a.blk_begin[i] = acc;
    for (int i = paths; i < 8; i++) a.dir_code[i] = 0;
}

// FSGM_AGG_FINE=0 keeps 16 costs a lane along x too (the round-1 split), for comparison
static bool agg_fine() {
    static const bool v = [] { const char* e = getenv("FSGM_AGG_FINE"); return !(e && e[0] == '0'); }();
    return v;
}

template <int D>
static void launch_packed(hipStream_t st, AggArgs& a, int paths, int frames, bool wrap) {
    const bool fine = agg_fine();
    constexpr int DO = AggSplit<D>::other, DX = AggSplit<D>::along_x;
    const bool lean = (D == 32 || D == 64 || D == 128) && !wrap && fine && FSGM_AGG_XLEAN;      // agg_x_lean_body: two costs a lane
    plan_dirs(a, paths, lean ? 4 * (128 / D) : 4 * (64 / (D / (fine ? DX : DO))), 4 * (64 / (D / DO)));
    dim3 grid(a.blk_begin[8], frames);
    if (wrap) {
        if (fine) hipLaunchKernelGGL((agg_packed_kernel<D, true, true>), grid, dim3(256), 0, st, a);
        else      hipLaunchKernelGGL((agg_packed_kernel<D, true, false>), grid, dim3(256), 0, st, a);
    } else {
        if (fine) hipLaunchKernelGGL((agg_packed_kernel<D, false, true>), grid, dim3(256), 0, st, a);
        else      hipLaunchKernelGGL((agg_packed_kernel<D, false, false>), grid, dim3(256), 0, st, a);
    }
}

void launch_aggregate(hipStream_t st, AggArgs a, int paths, int frames, int kernel_kind) {
    if (kernel_kind == AGG_GENERIC) {
        plan_dirs(a, paths, 4, 4);
        dim3 grid(a.blk_begin[8], frames);
        hipLaunchKernelGGL(agg_generic_kernel, grid, dim3(256), 0, st, a);
        return;
    }
    const bool wrap = kernel_kind == AGG_PACKED_WRAP;
    switch (a.D) {
        case 16: launch_packed<16>(st, a, paths, frames, wrap); break;
        case 32: launch_packed<32>(st, a, paths, frames, wrap); break;
        case 64: launch_packed<64>(st, a, paths, frames, wrap); break;
        case 128: launch_packed<128>(st, a, paths, frames, wrap); break;
        case 256: launch_packed<256>(st, a, paths, frames, wrap); break;
        default: break;
    }
}

void launch_wta(hipStream_t st, const WtaArgs& a, int frames, bool packed) {
    const int NP = a.W * a.H;
    if (packed) {
        const int lpp = agg_packed_lpp(a.D);
        dim3 grid((NP + 256 / lpp - 1) / (256 / lpp), frames);
        switch (lpp) {
            case 1: hipLaunchKernelGGL(wta_packed_kernel<1>, grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL(wta_packed_kernel<2>, grid, dim3(256), 0, st, a); break;
            case 4: hipLaunchKernelGGL(wta_packed_kernel<4>, grid, dim3(256), 0, st, a); break;
            case 8: hipLaunchKernelGGL(wta_packed_kernel<8>, grid, dim3(256), 0, st, a); break;
            case 16: hipLaunchKernelGGL(wta_packed_kernel<16>, grid, dim3(256), 0, st, a); break;
        }
    } else {
        dim3 grid((NP + 3) / 4, frames);
        hipLaunchKernelGGL(wta_generic_kernel, grid, dim3(256), 0, st, a);
    }
}

void launch_sum_paths(hipStream_t st, const uint8_t* L, uint32_t* S, size_t n, size_t dir_stride, int ndirs) {
    hipLaunchKernelGGL(sum_paths_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, L, S, n, dir_stride, ndirs);
}

}  // namespace fsgm
