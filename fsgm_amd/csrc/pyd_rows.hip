// pyd_rows.hip -- row-packed gfx950 kernels for the pyramidal 2-D variant
// (reference: calc_pyd_cost_sgm.cpp; citations per kernel).  Same results as the generic kernels
// of pyd_kernels.hip; used for search windows up to 11 x 11 (the reference's, pyramidal_sgm.m:17-19)
// and penalties in the no-wrap range.
//
// Layout ("rows", pyd_kernels.h): a pixel's candidates are Sx rows of RS = 4*NW bytes (Sy costs +
// padding), PS = Sx*RS bytes per pixel.  16 adjacent lanes (one DPP row) own one pixel, lane j the
// candidate row sx = j (cost, WTA) or sx = j-2 (aggregation); a wave works on 4 pixels / 4 path
// lines at a time, every lane moves its row as NW dwords.
#include "pyd_kernels.h"
#include "fsgm_device.h"
#include "epi_step.h"          // pk_min3: v_pk_minimum3_f16 as an exact u16 3-input minimum on values below 1024 (self-tested: capi_pyd.hip)
#include <type_traits>
#include <algorithm>

namespace fsgm {

constexpr int ROWS_MAXS = 11;           // largest window side of the rows layout
constexpr uint32_t ROWS_OUTSIDE = 0xFFFFFFFFu;   // a census code never has bit 0 set (common.cpp:21)

constexpr int DPP_ROW_SHL2 = 0x102;     // lane i <- lane i+2 (within a row of 16)
constexpr int DPP_ROW_SHR2 = 0x112;     // lane i <- lane i-2

typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
    u16x2 r = __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c);
    return __builtin_bit_cast(uint32_t, r);
}
// per half: 0xFFFF where idx < n, else 0   (idx2, n2 packed small non-negative values)
__device__ __forceinline__ uint32_t pk_lt_mask(uint32_t idx2, uint32_t n2) {
    i16x2 d = __builtin_bit_cast(i16x2, pk_sub(idx2, n2));
    d = d >> 15;
    return __builtin_bit_cast(uint32_t, d);
}
// (m & a) | (~m & b) as one v_bfi_b32; the empty asm keeps the mask a plain register (otherwise the
// compiler turns the selects back into compares on the halves)
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }
__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
// DPP row shift whose out-of-row lanes read 0: for results that are only used in lanes whose source
// lies inside the row (no register initialisation needed, unlike dpp_mov)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_shift0(uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t dup16(uint32_t v) { return v | (v << 16); }
// a candidate row (NW dwords, 4-byte aligned) as ONE memory instruction: the lanes of a pixel then cover its bytes without gaps
template <int N> struct alignas(4) RowDw { uint32_t v[N]; };
template <int N> __device__ __forceinline__ RowDw<N> load_row(const void* p) { return *(const RowDw<N>*)p; }
template <int N> __device__ __forceinline__ void store_row(void* p, const RowDw<N>& r) { *(RowDw<N>*)p = r; }

// =============================================================================================
// 2-D window cost  (calc_pyd_cost_sgm.cpp:374-437), aggregation radius <= 2.
// The sample column depends on offx+ax only and the sample row on offy+ay only (:415-416), so the
// (Sx+2r) x (Sy+2r) patch of image-2 census codes that all candidates and taps of a pixel touch is
// staged in LDS once; lane ox then walks its Sy candidates with the patch column in registers:
// (2r+1) x (Sy+2r) LDS reads per (2r+1)^2 x Sy taps, two VALU instructions per tap.  A wave with a
// sample outside image 2 (constant cost 5, USE_CONST_COST :32,405-421) takes a three-instruction
// form of the same loop (masked xor), one with a tap outside image 1 a four-instruction form.
// =============================================================================================
constexpr int COST_PM = ROWS_MAXS + 4;                   // max patch side
constexpr int COST_PPW = 8;                              // most pixels a wave takes
// a pixel's LDS slot (dwords): patch [PY][PX] | image-1 taps [AW][AW] | column table [16] | row table [16] | x, y, hint (6)
constexpr int COST_PATCH = 0, COST_C1 = COST_PM * COST_PM, COST_XT = COST_C1 + 25, COST_YT = COST_XT + 16, COST_HDR = COST_YT + 16;
constexpr int COST_SLOT = COST_HDR + 6 + 11;             // 288 + 11 dwords: lane l of the wave reads bank l + const in the tap
                                                         // loop of an 11-wide window (other widths: a few two-way conflicts)
constexpr uint32_t ROWS_OUTSIDE2 = 0x7C000000u;          // a sample outside image 2: five bits where no census code has any

// Sx lanes own a pixel (lane <-> candidate column ox), a wave takes 64 / Sx pixels (5 at the reference's 11 x 11 window:
// 55 of 64 lanes busy in the tap loop, against 44 with one pixel per DPP row).
__host__ __device__ inline int cost_ppw(int Sx) { return 64 / Sx < COST_PPW ? 64 / Sx : COST_PPW; }

// k / n as a 24-bit multiply and a shift: exact while k < 2048 and k * n < 2^20 (magic * n - 2^20 <= n); here k <= 8 * 15 * 15
// and n <= 15 * 15.  (The index splits of the fills were a third of the kernel's instructions as integer divisions.)
__host__ __device__ inline uint32_t div_magic(int n) { return (1u << 20) / (uint32_t)n + 1u; }
__device__ __forceinline__ int div_by(int k, uint32_t magic) { return (int)(__umul24((uint32_t)k, magic) >> 20); }
// base[idx] with a 32-bit index: the address stays "uniform base + 32-bit lane offset"
__device__ __forceinline__ uint32_t load_u32_at(const uint32_t* base, uint32_t idx) {
    return *(const uint32_t*)((const char*)base + (size_t)(idx * 4u));
}

template <int NW>
__global__ __launch_bounds__(256) void pyd_rows_cost_kernel(PydCostArgs a) {
    extern __shared__ uint32_t sCost[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = a.W, H = a.H, r = a.rAgg;
    const int NP = W * H;
    const int Sx = 2 * a.rX + 1, Sy = 2 * a.rY + 1;
    const int PX = Sx + 2 * r, PY = Sy + 2 * r, AW = 2 * r + 1, NA = AW * AW;
    const int ppw = cost_ppw(Sx);
    constexpr int SLOT = COST_SLOT;
    const uint32_t magicAW = div_magic(AW), magicNA = div_magic(NA);
    uint32_t* const wbase = sCost + wave * ppw * SLOT;
    const int sraw = div_by(lane, div_magic(Sx));
    const bool live = sraw < ppw;                          // lanes beyond the last whole pixel idle
    const int slot = live ? sraw : 0, j = live ? lane - sraw * Sx : 0;
    const int p0 = (blockIdx.x * 4 + wave) * ppw;          // the wave's first pixel
    const size_t f = blockIdx.y;
    const uint32_t* __restrict__ cen1 = a.cen1 + f * (size_t)NP;
    const uint32_t* __restrict__ cen2 = a.cen2 + f * (size_t)NP;
    if (lane < ppw) {                                      // pixel coordinates and hint, once per pixel
        const int p = min(p0 + lane, NP - 1);
        const int y = p / W, x = p - y * W;
        const double* mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
        const double* mvyp = mvxp + (size_t)a.mvW * a.mvH;
        uint32_t* h = wbase + lane * SLOT + COST_HDR;
        h[0] = (uint32_t)x;
        h[1] = (uint32_t)y;
        const double mvx = mvxp[(size_t)a.mvW * y + x], mvy = mvyp[(size_t)a.mvW * y + x];             // :388-389
        h[2] = (uint32_t)__double2loint(mvx); h[3] = (uint32_t)__double2hiint(mvx);
        h[4] = (uint32_t)__double2loint(mvy); h[5] = (uint32_t)__double2hiint(mvy);
    }
    __builtin_amdgcn_wave_barrier();
    bool outside1 = false;                                 // a tap outside image 1 / a sample outside image 2 (:405, :418)
    uint32_t outside2 = 0;
#pragma unroll
    for (int it = 0; it < COST_PPW * 16 / 64; it++) {      // sample column / row tables: entry k of pixel es
        const int e = lane + 64 * it, es = e >> 4, k = e & 15;
        if (es < ppw) {
            uint32_t* sl = wbase + es * SLOT;
            const int x = (int)sl[COST_HDR], y = (int)sl[COST_HDR + 1];
            const double mvx = __hiloint2double((int)sl[COST_HDR + 3], (int)sl[COST_HDR + 2]);
            const double mvy = __hiloint2double((int)sl[COST_HDR + 5], (int)sl[COST_HDR + 4]);
            // k = (offx + rX) + (ax + r)  ->  offx + x1 = x + k - rX - r
            const int x2 = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)(x + k - a.rX - r), mvx), 0.5));   // :416
            const int y2 = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)(y + k - a.rY - r), mvy), 0.5));   // :415
            sl[COST_XT + k] = (x2 >= 0 && x2 <= W - 1) ? (uint32_t)x2 : 0x80000000u;
            sl[COST_YT + k] = (y2 >= 0 && y2 <= H - 1) ? (uint32_t)(y2 * W) : 0x80000000u;
        }
    }
    // Fills: every address first, then all loads in flight together, then the LDS writes (as a plain loop each
    // iteration waited for its own gathered load).  Cells are dealt to the 64 lanes across the wave's pixels.
    {
        const int ncell = ppw * NA;                                  // (2r+1)^2 <= 25 image-1 taps per pixel
        uint32_t v[4];
        int dst[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int e = lane + 64 * it, ec = min(e, ncell - 1);
            const int es = div_by(ec, magicNA), kc = ec - es * NA;
            const int ky = div_by(kc, magicAW), kx = kc - ky * AW;
            const uint32_t* sl = wbase + es * SLOT;
            const int y1 = (int)sl[COST_HDR + 1] + ky - r, x1 = (int)sl[COST_HDR] + kx - r;
            const bool in = y1 >= 0 && y1 <= H - 1 && x1 >= 0 && x1 <= W - 1;                            // :405
            v[it] = load_u32_at(cen1, in ? (uint32_t)(W * y1 + x1) : 0u);
            if (!in) v[it] = ROWS_OUTSIDE;
            outside1 |= !in;
            dst[it] = e < ncell ? es * SLOT + COST_C1 + kc : -1;
        }
#pragma unroll
        for (int it = 0; it < 4; it++)
            if (dst[it] >= 0) wbase[dst[it]] = v[it];
    }
    __builtin_amdgcn_wave_barrier();
    {
        // patch of image-2 census codes: lane (kr, kx) = (lane / 16, lane % 16) takes the cells (kr + 4 * rr, kx) of one
        // pixel after the other -- no index arithmetic, the LDS offsets are immediates
        const int kx = lane & 15, kr = lane >> 4;
        int cell[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) cell[rr] = (kx < PX && kr + 4 * rr < PY) ? (kr + 4 * rr) * PX + kx : -1;
#pragma unroll
        for (int es = 0; es < COST_PPW; es++) {
            if (es >= ppw) continue;                                 // wave-uniform
            uint32_t* sl = wbase + es * SLOT;
            const uint32_t x2 = sl[COST_XT + kx];
            uint32_t v[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const uint32_t t = sl[COST_YT + kr + 4 * rr] | x2;   // sign bit: row or column outside
                const uint32_t yx = sl[COST_YT + kr + 4 * rr] + x2;
                const bool in = (int)t >= 0;                                                             // :418
                v[rr] = load_u32_at(cen2, in ? yx : 0u);
                if (!in) v[rr] = ROWS_OUTSIDE2;
                if (cell[rr] >= 0) outside2 |= t;
            }
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
                if (cell[rr] >= 0) sl[COST_PATCH + cell[rr]] = v[rr];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const bool any1 = __builtin_amdgcn_ballot_w64(outside1) != 0;             // wave-uniform
    const bool any2 = __builtin_amdgcn_ballot_w64((int)outside2 < 0) != 0;
    const uint32_t* const patch = wbase + slot * SLOT + COST_PATCH + j;        // [PY][PX], this lane's column ox = j
    const uint32_t* const c1s = wbase + slot * SLOT + COST_C1;                 // [AW][AW]
    uint32_t sum[4 * NW];
#pragma unroll
    for (int i = 0; i < 4 * NW; i++) sum[i] = 0;
    // The taps.  MODE 0: every sample inside both images, xor + popcount.  MODE 1: samples outside image 2 cost 5
    // (USE_CONST_COST :32,:419): their patch word is ROWS_OUTSIDE2 and the image-1 code is masked away before the xor.
    // MODE 2: image-1 taps outside as well (:406): those contribute nothing here and 5 per tap at the end.
    // Patch rows past PY and taps past AW read other words of the slot: they only reach sums that are dropped (oy >= Sy).
    auto taps = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        uint32_t extra = 0;
#pragma unroll
        for (int ax = 0; ax < 5; ax++) {
            if (ax >= AW) continue;
            uint32_t pc[4 * NW + 4], pm[MODE ? 4 * NW + 4 : 1], u[5];      // patch column ox+ax, rows 0 .. Sy+2r-1
#pragma unroll
            for (int q = 0; q < 4 * NW + 4; q++) pc[q] = patch[q * PX + ax];
#pragma unroll
            for (int ay = 0; ay < 5; ay++) u[ay] = c1s[ay * AW + ax];
            if (MODE) {
#pragma unroll
                for (int q = 0; q < 4 * NW + 4; q++) pm[q] = pc[q] == ROWS_OUTSIDE2 ? 0u : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int ay = 0; ay < 5; ay++) {
                if (ay >= AW) continue;
                uint32_t mu = 0xFFFFFFFFu;
                if (MODE == 2) {
                    const bool uin = u[ay] != ROWS_OUTSIDE;
                    mu = uin ? 0xFFFFFFFFu : 0u;
                    extra += uin ? 0u : 5u;
                }
#pragma unroll
                for (int oy = 0; oy < 4 * NW; oy++) {
                    if (MODE == 0)      sum[oy] += __popc(u[ay] ^ pc[oy + ay]);                          // :427
                    else if (MODE == 1) sum[oy] += __popc((u[ay] & pm[oy + ay]) ^ pc[oy + ay]);
                    else                sum[oy] += __popc(((u[ay] & pm[oy + ay]) ^ pc[oy + ay]) & mu);
                }
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int oy = 0; oy < 4 * NW; oy++) sum[oy] += extra;
        }
    };
    if (any1)      taps(std::integral_constant<int, 2>{});
    else if (any2) taps(std::integral_constant<int, 1>{});
    else           taps(std::integral_constant<int, 0>{});
    // (u8)(1.0*sum/win + 0.5) (:431) without the fp64 division: win = (2r+1)^2 is odd, so sum/win + 0.5
    // is never within 1/(2*win) of an integer and the truncation equals (2*sum + win) / (2*win) in
    // integers; that quotient by multiply-shift (exact for every sum up to 32*win, checked value by
    // value in tests/test_capi_cpu.py).
    const uint32_t win = (uint32_t)(AW * AW), inv = (1u << 20) / (2u * win) + 1u;
    uint32_t out[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int oy = 4 * k + b;
            const uint32_t c = oy < Sy ? (((2u * sum[oy] + win) * inv) >> 20) : 0u;
            w |= c << (8 * b);
        }
        out[k] = w;
    }
    if (live && p0 + slot < NP) {
        RowDw<NW> row;
#pragma unroll
        for (int k = 0; k < NW; k++) row.v[k] = out[k];
        store_row<NW>(a.C + (f * (size_t)NP + (size_t)(p0 + slot)) * a.PS + (size_t)j * a.RS, row);
    }
}

// =============================================================================================
// Per-step descriptors of the aggregation kernel: everything sgm_step derives from the hint map
// and the image rather than from the path state (calc_pyd_cost_sgm.cpp:46-47 hint shift, :91-95
// adaptive P2, :182 etc. path start).
//
// The shifted centre of candidate row sx is xpre = (int)(sx + dx + 0.5), a C truncation (:47):
// floor for non-negative arguments, ceil for negative ones.  Hence xpre(sx) = sx + kf + [sx < na]
// for some kf and a prefix length na -- two shift regimes -- for every real hint delta.  The
// descriptor stores (kf, na) per axis and a flag saying the table really has that form (checked
// entry by entry against the reference expression; entries whose 5x5 neighbourhood lies wholly
// outside the window may differ).  Steps without the flag take the aggregation kernel's gather path.
//   bits 0-5 kfx+16 | 6-9 nax | 10-15 kfy+16 | 16-19 nay | 20-27 P2 | 28 path start | 29 model ok
// =============================================================================================
constexpr uint32_t DESC_START = 1u << 28, DESC_OK = 1u << 29;

__device__ __forceinline__ bool absent_centre(int c, int S) { return c <= -3 || c >= S + 2; }

__device__ bool fit_shift(double delta, int S, int& kf, int& na) {
    int t[ROWS_MAXS];
    int jstar = -1, tstar = 0;
    // (|delta| < 2^30: every j + delta + 0.5 lies inside the int range, where cvttsd2si is the plain truncating conversion)
    if (fabs(delta) < 1073741824.0) {
#pragma unroll
        for (int j = 0; j < ROWS_MAXS; j++) t[j] = (int)__dadd_rn(__dadd_rn((double)j, delta), 0.5);
    } else {
#pragma unroll
        for (int j = 0; j < ROWS_MAXS; j++) t[j] = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)j, delta), 0.5));
    }
#pragma unroll
    for (int j = 0; j < ROWS_MAXS; j++)
        if (j < S && !absent_centre(t[j], S)) { jstar = j; tstar = t[j]; }
    if (jstar < 0) { kf = S + 3; na = 0; return true; }      // every neighbourhood lies outside the window
    kf = tstar - jstar;
    na = 0;
#pragma unroll
    for (int j = 0; j < ROWS_MAXS; j++)
        if (j < jstar && t[j] != j + kf) na = j + 1;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < ROWS_MAXS; j++)
        if (j < S) {
            const int model = j + kf + (j < na ? 1 : 0);
            ok = ok && (t[j] == model || (absent_centre(t[j], S) && absent_centre(model, S)));
        }
    return ok;
}

__global__ __launch_bounds__(256) void pyd_rows_desc_kernel(PydAggArgs a) {
    const int W = a.W, H = a.H, NP = W * H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const int slot = blockIdx.y;
    const size_t f = blockIdx.z;
    const int code = a.dir_code[slot], base = code & 3;
    const bool mirror = (code & 4) != 0;
    const int rx = base == 1 ? 0 : (base == 3 ? -1 : 1), ry = base == 0 ? 0 : 1;
    const int ay = p / W, ax = p - ay * W;
    const int px = mirror ? ax + rx : ax - rx, py = mirror ? ay + ry : ay - ry;   // path predecessor
    uint32_t d;
    if (px < 0 || px >= W || py < 0 || py >= H) {
        d = DESC_START | DESC_OK | (16u) | (16u << 10);
    } else {
        const double* mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
        const double* mvyp = mvxp + (size_t)a.mvW * a.mvH;
        const double dx = __dsub_rn(mvxp[(size_t)ay * a.mvW + ax], mvxp[(size_t)py * a.mvW + px]);   // :213 etc.
        const double dy = __dsub_rn(mvyp[(size_t)ay * a.mvW + ax], mvyp[(size_t)py * a.mvW + px]);
        const uint8_t* If = a.I1 + f * (size_t)NP;
        const int dI = abs((int)If[(size_t)W * ay + ax] - (int)If[(size_t)W * py + px]);
        const int P2 = (a.adaptive && dI > 50) ? a.P2 / 8 : a.P2;                                    // :91-95
        int kfx, nax, kfy, nay;
        const bool okx = fit_shift(dx, a.Sx, kfx, nax);
        const bool oky = fit_shift(dy, a.Sy, kfy, nay);
        d = (uint32_t)(kfx + 16) | ((uint32_t)nax << 6) | ((uint32_t)(kfy + 16) << 10) | ((uint32_t)nay << 16) |
            ((uint32_t)P2 << 20) | ((okx && oky) ? DESC_OK : 0u);
    }
    a.desc[(f * a.ndirs + slot) * (size_t)NP + p] = d;
}

// =============================================================================================
// 2-D path aggregation  (calc_pyd_cost_sgm.cpp:34-89 sgm_step, :114-296 sgm2d), no-wrap penalties
// (0 <= P1,P2, max C + P2 + max(P1,P2) <= 255: no u8 narrowing changes a value, so the centre cell
// may take part in the "+P1" minimum and a cell outside the window is just 255).
//
// The previous pixel's path costs sit in LDS as bytes, one 52-byte row per candidate row with the
// Sy costs at byte 16 and 0xFF around them (row 15 is all 0xFF).  Lane j of a 16-lane DPP row reads
// candidate row (j-2)+kfx at byte offset kfy-2 -- the hint shift (:46-47) folded into the address --
// realigns the dwords with v_alignbyte, and takes the 5-wide minimum along sy in packed u16
// registers (even/odd split, as in the 1-D kernels); the 5-wide minimum along sx is 4 DPP row
// shifts.  The second shift regime of a truncating conversion (pyd_rows_desc_kernel) is a
// per-element select along sy and a one-lane DPP shift along sx.  Steps whose shift table is
// irregular gather the 5x5 cells byte by byte.
//
// Two mappings of the same step:
//   packed (WIDE = false): a wave advances 4 path lines, one per DPP row; a lane owns a whole
//          candidate row (NW dwords).  Least work per pixel: used when there are many lines.
//   wide   (WIDE = true):  a wave advances ONE line; DPP row q owns the sy quarter 4q..4q+3 of every
//          candidate row (1 dword per lane).  2.5x fewer instructions per step: used for the long
//          horizontal lines of a single frame, whose serial length bounds the whole stage.
// =============================================================================================
constexpr int ROWB = 52;                // bytes per LDS row (13 dwords: odd stride)
constexpr int ROWDATA = 16;             // byte of sy = 0
constexpr int ABSROW = 15;              // the all-0xFF row
constexpr int DUMPROW = 14;             // written by lanes without a candidate row, never read
constexpr int ROWS_WAVE_BYTES = 2 * 4 * 16 * ROWB + 4 * 32 * 4;    // two buffers of 4 x 16 rows + gather tables

template <int NW, bool WIDE>
__device__ __forceinline__ void pyd_rows_agg_body(const PydAggArgs& a, const int slot, uint32_t* sRows) {
#ifndef FSGM_PYD_PF
#define FSGM_PYD_PF 4
#endif
    constexpr int PF = FSGM_PYD_PF;                          // prefetch distance in steps (A/B knob: 6 and 8 measured no better)
    constexpr int NL = WIDE ? 1 : NW;                        // dwords of a candidate row this lane owns
    constexpr int NE = NL + (WIDE ? 2 : 1);                  // realigned dwords of the previous row it needs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, row4 = lane >> 4;
    const int code = a.dir_code[slot];
    const int base = WIDE ? 0 : (code & 3);                 // (the one-line-per-wave mapping is only ever given the along-x slots: the walk below folds to +-1)
    const bool mirror = (code & 4) != 0;
    const int W = a.W, H = a.H, Sx = a.Sx, Sy = a.Sy, RS = a.RS, PS = a.PS;
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    int line = ((int)blockIdx.x - a.blk_begin[slot]) * (WIDE ? 4 : 16) + (WIDE ? wave : wave * 4 + row4);
    const bool active = line < nlines;
    line = min(line, nlines - 1);
    const int q = WIDE ? min(row4, NW - 1) : 0;              // sy quarter (wide); rows beyond the last quarter idle
    const bool qlive = !WIDE || row4 < NW;
    uint8_t* const wbase = (uint8_t*)sRows + (size_t)wave * ROWS_WAVE_BYTES;
    for (int i = lane; i < 2 * 4 * 16 * ROWB / 4; i += 64) ((uint32_t*)wbase)[i] = 0xFFFFFFFFu;
    uint8_t* pre = wbase + (WIDE ? 0 : row4) * 16 * ROWB;
    uint8_t* cur = pre + 4 * 16 * ROWB;
    int32_t* const tabs = (int32_t*)(wbase + 2 * 4 * 16 * ROWB) + (WIDE ? 0 : row4) * 32;   // gather path: xtab[16], ytab[16]
    const size_t f = blockIdx.y;
    const uint8_t* __restrict__ Cf = a.C + f * (size_t)NP * PS;
    uint8_t* __restrict__ Lf = a.L + (f * a.ndirs + slot) * (size_t)NP * PS;
    const uint32_t* __restrict__ Df = a.desc + (f * a.ndirs + slot) * (size_t)NP;
    const int sx = j - 2;                                    // candidate row this lane produces
    const bool is_out = sx >= 0 && sx < Sx && qlive;
    const int sxc = clampi(sx, 0, Sx - 1);
    const uint32_t rowoff = (uint32_t)(sxc * RS + 4 * q);    // this lane's bytes inside a pixel
    // padding slots (sy >= Sy) carry 0x00FF so that the row written back reads as "outside"; lanes
    // that produce nothing carry all ones so that they never win the minimum
    uint32_t padE[NL], padO[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int s0 = 4 * (k + q);
        padE[k] = is_out ? ((s0 >= Sy ? 0x00FFu : 0u) | (s0 + 2 >= Sy ? 0x00FF0000u : 0u)) : 0xFFFFFFFFu;
        padO[k] = is_out ? ((s0 + 1 >= Sy ? 0x00FFu : 0u) | (s0 + 3 >= Sy ? 0x00FF0000u : 0u)) : 0xFFFFFFFFu;
    }
    const uint32_t P1_2 = dup16((uint32_t)a.P1);

    // Walk of the line as increments of the pixel index.  Pass-0 frame: step (+1,0), (0,+1), (+1,+1)
    // or (-1,+1); a diagonal that leaves the image re-enters at the opposite border (a path start),
    // so every line has `len` steps.  The mirrored passes run the same walk through the point mirror.
    const int sgn = mirror ? -1 : 1;
    const int dstep = sgn * (base == 0 ? 1 : base == 1 ? W : base == 2 ? W + 1 : W - 1);
    const int dwrap = sgn * (base == 2 ? -W : W);            // extra increment when the diagonal wraps
    const int dcx = base == 3 ? -1 : 1;
    const int cwrap = base == 2 ? W : -1, creset = base == 2 ? 0 : W - 1;
    int cx = base == 0 ? 0 : line;                           // pass-0 column (diagonals only)
    int pix = base == 0 ? line * W : line;                   // first pixel: (0, line) for rows, (line, 0) otherwise
    if (mirror) pix = NP - 1 - pix;                          // (shadowed inside the loop by the current step's pixel)
    // (the byte offset of the pixel's bytes in a volume moves along with the index: no multiplication, no 64-bit address per step --
    //  a frame's volume is below 4 GB or the plan has no descriptors and this kernel is not used, capi_pyd.hip)
    const uint32_t dstepB = (uint32_t)(dstep * PS), dwrapB = (uint32_t)(dwrap * PS);
    auto advance = [&](int& p, int& c, uint32_t& off) {
        p += dstep;
        off += dstepB;
        if (base >= 2) {
            c += dcx;
            if (c == cwrap) { c = creset; p += dwrap; off += dwrapB; }
        }
    };
    // Prefetch ring: costs and descriptor of step t+PF are requested while step t computes (HBM latency
    // is several steps long).  The ring rotates by unrolling, not by copying: a copy would wait for
    // the load it copies.  The fetch cursor runs PF steps ahead of the step being computed; steps past
    // the end of the line re-read the first pixel and store into the dump slot.
    struct Fetch { uint32_t c[NL]; uint32_t desc; int pix; uint32_t off; };
    const int pix0 = pix;
    const uint32_t off0 = (uint32_t)pix * (uint32_t)PS + rowoff;   // this lane's bytes of the line's first pixel
    int fpix = pix, fcx = cx, ft = 0;
    uint32_t foff = off0;
    auto issue = [&](Fetch& o) {
        const bool in_line = ft < len;
        const int p = in_line ? fpix : pix0;
        const uint32_t off = in_line ? foff : off0;
        const RowDw<NL> row = load_row<NL>((const uint8_t*)Cf + off);
#pragma unroll
        for (int k = 0; k < NL; k++) o.c[k] = row.v[k];
        o.desc = *(const uint32_t*)((const uint8_t*)Df + ((uint32_t)p << 2));
        o.pix = p;
        o.off = off;
        ft++;
        advance(fpix, fcx, foff);
    };
    Fetch ring[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) issue(ring[k]);
    uint32_t m = 0;
    __builtin_amdgcn_wave_barrier();
    for (int t0 = 0; t0 < len; t0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int t = t0 + u;
        const Fetch now = ring[u];
        const int pix = now.pix;
        issue(ring[u]);                                      // step t+PF, in flight while steps t .. t+PF-1 compute
        const uint32_t d = now.desc;
        const bool start = (d & DESC_START) != 0;
        const int kfx = (int)(d & 63u) - 16, nax = (int)((d >> 6) & 15u);
        const int kfy = (int)((d >> 10) & 63u) - 16, nay = (int)((d >> 16) & 15u);
        const uint32_t P2 = (d >> 20) & 255u;
        uint32_t ME[NL], MO[NL], CE[NL], CO[NL];            // 5x5 minimum / centre value per sy (even, odd sy)
        const bool gather = __builtin_amdgcn_ballot_w64((d & DESC_OK) == 0) != 0;
        if (!gather) {
            const int R = sx + kfx;
            const int rsel = (R >= 0 && R < Sx) ? R : ABSROW;
            const int b0 = ROWDATA - 2 + kfy + 4 * q;
            const uint32_t* qp = (const uint32_t*)(pre + rsel * ROWB + (b0 & ~3));
            uint32_t qd[NE + 1];
#pragma unroll
            for (int i = 0; i < NE + 1; i++) qd[i] = qp[i];
            const uint32_t sh = (uint32_t)b0 & 3u;
            // v[i] = Lpre(R, kfy-2+4q+i): E[k] = (v[4k], v[4k+2]), O[k] = (v[4k+1], v[4k+3])
            uint32_t E[NE], O[NE];
#pragma unroll
            for (int k = 0; k < NE; k++) {
                const uint32_t w = __builtin_amdgcn_alignbyte(qd[k + 1], qd[k], sh);
                E[k] = w & 0x00FF00FFu;
                O[k] = (w >> 8) & 0x00FF00FFu;
            }
            uint32_t A[NL + 1], B[NL + 1], Es[NL], Os[NL];  // pair minima a2[i] = min(v[i], v[i+1]); shifted views
#pragma unroll
            for (int k = 0; k <= NL; k++) A[k] = pk_min(E[k], O[k]);
#pragma unroll
            for (int k = 0; k < NL; k++) {
                Es[k] = align16(E[k + 1], E[k]);            // (v[4k+2], v[4k+4])
                Os[k] = align16(O[k + 1], O[k]);            // (v[4k+3], v[4k+5])
                B[k] = pk_min(O[k], Es[k]);
            }
            B[NL] = pk_min(O[NL], align16(E[NL], E[NL]));   // only its low half is used
            uint32_t A5[NL + 1], B5[NL];                    // a5[i] = min(v[i .. i+4]) at even / odd i
#pragma unroll
            for (int k = 0; k < NL; k++) {
                A5[k] = pk_min3(A[k], align16(A[k + 1], A[k]), E[k + 1]);       // (all operands are bytes: below 1024)
                B5[k] = pk_min3(B[k], align16(B[k + 1], B[k]), O[k + 1]);
            }
            // a5[4*NL] (low half): the second regime's last odd element.  The packed mapping never
            // consumes it (that slot is padding); the wide one does.
            A5[NL] = WIDE ? pk_min3(A[NL], align16(A[NL], A[NL]), E[NE - 1]) : A5[NL - 1];
            // regime B: centre sy + kfy -> window v[sy .. sy+4], centre v[sy+2]
#pragma unroll
            for (int k = 0; k < NL; k++) { ME[k] = A5[k]; MO[k] = B5[k]; CE[k] = Es[k]; CO[k] = Os[k]; }
            if (__builtin_amdgcn_ballot_w64(nay != 0) != 0) {
                // regime A (sy < nay): centre one further, window v[sy+1 .. sy+5], centre v[sy+3]
                const uint32_t n2 = dup16((uint32_t)nay);
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    const uint32_t s0 = (uint32_t)(4 * (k + q));
                    const uint32_t mE = opaque(pk_lt_mask(s0 | ((s0 + 2) << 16), n2));
                    const uint32_t mO = opaque(pk_lt_mask((s0 + 1) | ((s0 + 3) << 16), n2));
                    const uint32_t A5s = align16(A5[k + 1], A5[k]);              // (a5[4k+2], a5[4k+4])
                    ME[k] = bfi(mE, B5[k], A5[k]);
                    MO[k] = bfi(mO, A5s, B5[k]);
                    CE[k] = bfi(mE, Os[k], Es[k]);
                    CO[k] = bfi(mO, E[k + 1], Os[k]);
                }
            }
            // 5-wide minimum along sx: lanes j-2 .. j+2 hold rows R-2 .. R+2.  Only lanes 2 .. 13 are
            // consumed (producing lanes 2 .. Sx+1 <= 12, plus one for the second regime), and all their
            // sources lie inside the DPP row.
#pragma unroll
            for (int k = 0; k < 2 * NL; k++) {
                const uint32_t v = k < NL ? ME[k] : MO[k - NL];
                const uint32_t a2 = pk_min(v, dpp_shift0<DPP_ROW_SHL1>(v));         // lanes j, j+1
                const uint32_t h = pk_min3(dpp_shift0<DPP_ROW_SHR2>(a2), a2, dpp_shift0<DPP_ROW_SHL2>(v));   // j-2, j-1 | j, j+1 | j+2
                if (k < NL) ME[k] = h; else MO[k - NL] = h;
            }
            if (__builtin_amdgcn_ballot_w64(nax != 0) != 0) {
                const bool regA = sx < nax;                  // centre row one further: take lane j+1's values
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    const uint32_t a1 = dpp_shift0<DPP_ROW_SHL1>(ME[k]), a2 = dpp_shift0<DPP_ROW_SHL1>(MO[k]);
                    const uint32_t a3 = dpp_shift0<DPP_ROW_SHL1>(CE[k]), a4 = dpp_shift0<DPP_ROW_SHL1>(CO[k]);
                    ME[k] = regA ? a1 : ME[k];
                    MO[k] = regA ? a2 : MO[k];
                    CE[k] = regA ? a3 : CE[k];
                    CO[k] = regA ? a4 : CO[k];
                }
            }
        } else {
            // gather path: the reference's tables (:46-47), 25 cells per candidate
            const int rx = base == 1 ? 0 : (base == 3 ? -1 : 1), ry = base == 0 ? 0 : 1;
            const int ay = pix / W, ax = pix - ay * W;
            const int px = clampi(mirror ? ax + rx : ax - rx, 0, W - 1), py = clampi(mirror ? ay + ry : ay - ry, 0, H - 1);
            const double* __restrict__ mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
            const double* __restrict__ mvyp = mvxp + (size_t)a.mvW * a.mvH;
            const double dx = __dsub_rn(mvxp[(size_t)ay * a.mvW + ax], mvxp[(size_t)py * a.mvW + px]);
            const double dy = __dsub_rn(mvyp[(size_t)ay * a.mvW + ax], mvyp[(size_t)py * a.mvW + px]);
            tabs[j] = clampi(f64_to_i32_x86(__dadd_rn(__dadd_rn((double)j, dx), 0.5)), -3, Sx + 2);        // :47
            tabs[16 + j] = clampi(f64_to_i32_x86(__dadd_rn(__dadd_rn((double)j, dy), 0.5)), -3, Sy + 2);   // :46
            __builtin_amdgcn_wave_barrier();
            const int cxr = tabs[sxc];
            auto rowp = [&](int tx) { return pre + ((tx >= 0 && tx < Sx) ? tx : ABSROW) * ROWB + ROWDATA; };
#pragma unroll
            for (int k = 0; k < NL; k++) { ME[k] = MO[k] = CE[k] = CO[k] = 0x00FF00FFu; }
#pragma unroll
            for (int s = 0; s < 4 * NL; s++) {
                uint32_t ctr = 255, nb = 255;
                const int sy = s + 4 * q;
                if (sy < Sy) {
                    const int cy = tabs[16 + sy];
                    ctr = rowp(cxr)[cy];
#pragma unroll 1
                    for (int mm = -2; mm <= 2; mm++) {
                        const uint8_t* rp = rowp(cxr + mm) + cy;
                        nb = min(nb, min(min((uint32_t)rp[-2], (uint32_t)rp[-1]), min(min((uint32_t)rp[0], (uint32_t)rp[1]), (uint32_t)rp[2])));
                    }
                }
                const int k = s >> 2;
                const uint32_t shv = (s & 2) ? 16 : 0, keep = (s & 2) ? 0x0000FFFFu : 0xFFFF0000u;
                if (s & 1) { MO[k] = (MO[k] & keep) | (nb << shv); CO[k] = (CO[k] & keep) | (ctr << shv); }
                else       { ME[k] = (ME[k] & keep) | (nb << shv); CE[k] = (CE[k] & keep) | (ctr << shv); }
            }
        }
        // best = min(m + P2, centre, 5x5 minimum + P1) (:50-80); L = C + best - m (:83).  At a path
        // start L = C (:153 etc.): best and m forced to 0 (every other operand of the minimum is >= 0).
        const uint32_t jump2 = start ? 0u : dup16(m + P2), m2 = start ? 0u : dup16(m);
        const uint32_t negm2 = 0u - m2;                      // C + best - m as one 32-bit three-input add: no half of the result is negative, so no borrow crosses
        uint32_t LE[NL], LO[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const uint32_t cE = now.c[k] & 0x00FF00FFu, cO = (now.c[k] >> 8) & 0x00FF00FFu;
            const uint32_t bE = pk_min3(jump2, CE[k], pk_add(ME[k], P1_2));      // m + P2, centre, minimum + P1: each below 2 * 255 + 1
            const uint32_t bO = pk_min3(jump2, CO[k], pk_add(MO[k], P1_2));
            LE[k] = (cE + bE + negm2) | padE[k];
            LO[k] = (cO + bO + negm2) | padO[k];
        }
        uint32_t rmin = pk_min(LE[0], LO[0]);
#pragma unroll
        for (int k = 1; k < NL; k++) rmin = pk_min(rmin, pk_min(LE[k], LO[k]));
        uint32_t lo = min(rmin & 0xFFFFu, rmin >> 16);
        lo = group_min_u32<16>(lo);
        if (WIDE) {                                          // one line per wave: the minimum is wave-uniform
            const uint32_t l0 = __builtin_amdgcn_readlane(lo, 0), l1 = __builtin_amdgcn_readlane(lo, 16);
            const uint32_t l2 = __builtin_amdgcn_readlane(lo, 32), l3 = __builtin_amdgcn_readlane(lo, 48);
            lo = min(min(l0, l1), min(l2, l3));
        }
        m = start ? 0u : lo;                                 // :88; stored minimum 0 at a start (:154)
        {
            // Straight-line stores: lanes that produce nothing write LDS row 14 (never read) and a dump
            // slot in HBM.  With one store per step at a fixed place in the instruction stream the
            // wait for the prefetched loads is a counted s_waitcnt vmcnt(1), not a wait for the store.
            uint32_t* dl = (uint32_t*)(cur + (is_out ? sx : DUMPROW) * ROWB + ROWDATA + 4 * q);
            uint32_t* dg = (is_out && active && t < len) ? (uint32_t*)(Lf + now.off) : a.dump;
            RowDw<NL> row;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                row.v[k] = LE[k] | (LO[k] << 8);
                dl[k] = row.v[k];
            }
            store_row<NL>(dg, row);
        }
        __builtin_amdgcn_wave_barrier();
        uint8_t* tmp = pre; pre = cur; cur = tmp;
      }
    }
}

template <int NW>
__global__ __launch_bounds__(256) void pyd_rows_agg_kernel(PydAggArgs a) {
    extern __shared__ uint32_t sRows[];
    int slot = 0;
#pragma unroll
    for (int i = 1; i < 8; i++)
        if (i < a.ndirs && (int)blockIdx.x >= a.blk_begin[i]) slot = i;
    if ((a.wide_mask >> slot) & 1) pyd_rows_agg_body<NW, true>(a, slot, sRows);
    else                           pyd_rows_agg_body<NW, false>(a, slot, sRows);
}

// =============================================================================================
// WTA + y/x parabola  (calc_pyd_cost_sgm.cpp:298-364).  4 pixels per wave; lane j sums the path
// costs of candidate row j (byte dot products over the directions), the first minimum in candidate
// order is the minimum of (sum << 8 | index) keys.
// =============================================================================================
template <int NW>
__global__ __launch_bounds__(256) void pyd_rows_wta_kernel(PydWtaArgs a) {
    constexpr int PIXDW = 16 * 2 * NW + 1;                 // LDS dwords per pixel: 16 candidate rows of 4*NW u16 sums; odd, so
    __shared__ uint32_t sSum[4 * 4 * PIXDW];               // that the wave's four pixels sit in different banks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, slot4 = lane >> 4;
    const int NP = a.W * a.H, Sy = a.Sy, Sx = a.Sx, D = Sx * Sy;
    int p = (blockIdx.x * 4 + wave) * 4 + slot4;
    const bool active = p < NP;
    p = min(p, NP - 1);
    const size_t f = blockIdx.y;
    const size_t vol = (size_t)NP * a.PS;
    const int sxc = min(j, Sx - 1);
    const uint8_t* __restrict__ Lp = a.L + f * a.ndirs * vol + (size_t)p * a.PS + (size_t)sxc * a.RS;
    uint32_t v[8][NW];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const RowDw<NW> row = load_row<NW>(Lp + (r < a.ndirs ? r : 0) * vol);      // (a direction that is not there: weight 0)
#pragma unroll
        for (int k = 0; k < NW; k++) v[r][k] = row.v[k];
    }
    // Sp += L (:227-232) as byte dot products: the four bytes of a dword are four sy, so the dwords of four directions are
    // transposed (8 v_perm) into one dword per sy holding that sy of the four directions, and v_dot4_u32_u8 with the
    // directions' weights as bytes (a third pass repeats the mirrored one) adds them up: 3 instructions per loaded dword.
    const uint32_t wlo = a.weight[0] | (a.weight[1] << 8) | (a.weight[2] << 16) | (a.weight[3] << 24);
    const uint32_t whi = a.weight[4] | (a.weight[5] << 8) | (a.weight[6] << 16) | (a.weight[7] << 24);
    uint32_t S[4 * NW];
#pragma unroll
    for (int k = 0; k < NW; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) S[4 * k + b] = 0;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const uint32_t d0 = v[4 * g][k], d1 = v[4 * g + 1][k], d2 = v[4 * g + 2][k], d3 = v[4 * g + 3][k];
            const uint32_t lo01 = __builtin_amdgcn_perm(d1, d0, 0x05010400u), hi01 = __builtin_amdgcn_perm(d1, d0, 0x07030602u);
            const uint32_t lo23 = __builtin_amdgcn_perm(d3, d2, 0x05010400u), hi23 = __builtin_amdgcn_perm(d3, d2, 0x07030602u);
            const uint32_t c0 = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u), c1 = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
            const uint32_t c2 = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u), c3 = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
            const uint32_t w = g ? whi : wlo;
            S[4 * k + 0] = __builtin_amdgcn_udot4(c0, w, S[4 * k + 0], false);
            S[4 * k + 1] = __builtin_amdgcn_udot4(c1, w, S[4 * k + 1], false);
            S[4 * k + 2] = __builtin_amdgcn_udot4(c2, w, S[4 * k + 2], false);
            S[4 * k + 3] = __builtin_amdgcn_udot4(c3, w, S[4 * k + 3], false);
        }
    }
    uint32_t* const pixS = sSum + (wave * 4 + slot4) * PIXDW;
#pragma unroll
    for (int i = 0; i < 2 * NW; i++) pixS[j * 2 * NW + i] = S[2 * i] | (S[2 * i + 1] << 16);
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int s = 0; s < 4 * NW; s++) {
        if (s < Sy && j < Sx) {
            key = min(key, (S[s] << 8) | (uint32_t)(j * Sy + s));                // :302-311 first strict minimum
            if (a.S && active) a.S[(f * NP + p) * (size_t)D + j * Sy + s] = S[s];
        }
    }
    key = group_min_u32<16>(key);
    __builtin_amdgcn_wave_barrier();
    if (j < 2 && active) {                                   // lane 0: outputs and the y parabola, lane 1: the x parabola
        const uint32_t gidx = key & 0xFFu, glo = key >> 8;
        if (j == 0) {
            a.bestD[f * NP + p] = gidx;
            a.minC[f * NP + p] = glo;
        }
        const int dx = (int)((gidx * ((1u << 20) / (uint32_t)Sy + 1u)) >> 20), dy = (int)gidx - dx * Sy;   // :333-334 (gidx / Sy)
        const uint16_t* Sp = (const uint16_t*)pixS;          // [16][4 * NW]
        const int at = dx * 4 * NW + dy, stride = j == 0 ? 1 : 4 * NW;
        const bool inner = j == 0 ? (dy > 0 && dy < Sy - 1) : (dx > 0 && dx < Sx - 1);
        double sub = 0.0;                                    // zero when subpixel is off (:476 zero-init output)
        if (a.subpixel && inner) sub = pyd_parabola((double)Sp[at - stride], (double)glo, (double)Sp[at + stride]);
        a.mvSub[f * 2 * (size_t)NP + (j == 0 ? NP : 0) + p] = sub;              // plane 0: x, plane 1: y
    }
}

// =============================================================================================
// launchers
// =============================================================================================
bool pyd_rows_cost_ok(const PydCostArgs& a) {
    const int Sx = 2 * a.rX + 1, Sy = 2 * a.rY + 1;
    return pyd_rows_layout(Sx, Sy) && a.RS == pyd_row_stride(Sx, Sy) && a.rAgg <= 2 && (long long)a.W * a.H < (1LL << 30);
}

bool pyd_rows_wta_ok(const PydWtaArgs& a) {
    uint32_t wsum = 0;
    for (int r = 0; r < a.ndirs; r++) {
        wsum += a.weight[r];
        if (a.weight[r] > 255u) return false;
    }
    return pyd_rows_layout(a.Sx, a.Sy) && a.RS == pyd_row_stride(a.Sx, a.Sy) && wsum * 255u <= 65535u;   // weights are bytes, sums u16
}

#define FSGM_ROWS_DISPATCH(NWV, CALL)            \
    do {                                         \
        if ((NWV) == 1) { constexpr int NW = 1; CALL; } \
        else if ((NWV) == 2) { constexpr int NW = 2; CALL; } \
        else { constexpr int NW = 3; CALL; }     \
    } while (0)

void launch_pyd_rows_cost(hipStream_t st, const PydCostArgs& a, int frames) {
    const int Sx = 2 * a.rX + 1, per_block = 4 * cost_ppw(Sx);
    dim3 grid((unsigned)((a.W * a.H + per_block - 1) / per_block), frames);
    const size_t lds = (size_t)4 * cost_ppw(Sx) * COST_SLOT * sizeof(uint32_t);
    FSGM_ROWS_DISPATCH(a.RS / 4, hipLaunchKernelGGL((pyd_rows_cost_kernel<NW>), grid, dim3(256), lds, st, a));
}

void launch_pyd_rows_desc(hipStream_t st, const PydAggArgs& a, int frames) {
    dim3 grid((unsigned)((a.W * a.H + 255) / 256), a.ndirs, frames);
    hipLaunchKernelGGL(pyd_rows_desc_kernel, grid, dim3(256), 0, st, a);
}

void launch_pyd_rows_aggregate(hipStream_t st, const PydAggArgs& a, int frames) {
    if (a.ndirs == 0) return;
    dim3 grid(a.blk_begin[8], frames);
    const size_t lds = (size_t)4 * ROWS_WAVE_BYTES;
    FSGM_ROWS_DISPATCH(a.RS / 4, hipLaunchKernelGGL((pyd_rows_agg_kernel<NW>), grid, dim3(256), lds, st, a));
}

void launch_pyd_rows_wta(hipStream_t st, const PydWtaArgs& a, int frames) {
    dim3 grid((unsigned)((a.W * a.H + 15) / 16), frames);
    FSGM_ROWS_DISPATCH(a.RS / 4, hipLaunchKernelGGL((pyd_rows_wta_kernel<NW>), grid, dim3(256), 0, st, a));
}

}  // namespace fsgm
