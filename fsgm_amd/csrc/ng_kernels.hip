// ng_kernels.hip -- gfx950 kernels for the two neighbour-guided variants.
//   calc_pyd_cost_sgm_ng.cpp : 9 hints x (2r+1)^2 candidates per pixel as {mvx,mvy,cost}, O(D^2)
//                              matcher per path step, 2 passes x 2 paths.
//   calc_cost_sgm_ng.cpp     : candidates come from the path buffers of pixels already visited
//                              (+ a rand() hint), so the frame is inherently raster-serial.
#include "ng_kernels.h"
#include "fsgm_device.h"
#include <stdlib.h>

#ifndef FSGM_NG_PRIO
#define FSGM_NG_PRIO 1
#endif
#ifndef FSGM_NG_CPF
#define FSGM_NG_CPF 8      // compact kernel: steps of list entries in flight (4: a single 1242x375 pair 1.67 ms, 8: 1.56, 2: 1.84)
#endif
namespace fsgm {

__device__ __forceinline__ bool near2(int a, int b) {
    const long long d = (long long)a - b;                    // abs(int - int) <= 2 without overflow
    return d >= -2 && d <= 2;
}

// =============================================================================================
// candidate list + cost  (calc_pyd_cost_sgm_ng.cpp:370-446).  One thread = one (pixel, candidate).
// Hint order: dy outer, dx inner over {-8,0,+8} (clamped to the hint map, :390-396); expansion:
// offx outer, offy inner (:399-400); sample = (int)((off + p1) + mv), no +0.5 (:417-418).
// =============================================================================================
__global__ __launch_bounds__(256) void ng_cost_kernel(NgCostArgs a) {
    const int W = a.W, H = a.H;
    const int NP = W * H;
    const int Sy = 2 * a.rY + 1, cph = (2 * a.rX + 1) * Sy, D = 9 * cph;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)NP * D) return;
    const int p = (int)(gid / D), d = (int)(gid - (long long)p * D);
    const int y = p / W, x = p - y * W;
    const int h = d / cph, c = d - h * cph;
    const int dy = (h / 3 - 1) * 8, dx = (h % 3 - 1) * 8;
    const int offx = c / Sy - a.rX, offy = c % Sy - a.rY;
    const size_t f = blockIdx.y;
    const double* mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
    const double* mvyp = mvxp + (size_t)a.mvW * a.mvH;
    const int yn = clampi(y + dy, 0, a.mvH - 1), xn = clampi(x + dx, 0, a.mvW - 1);
    const double mvx = mvxp[(size_t)a.mvW * yn + xn], mvy = mvyp[(size_t)a.mvW * yn + xn];
    const uint32_t* cen1 = a.cen1 + f * (size_t)NP;
    const uint32_t* cen2 = a.cen2 + f * (size_t)NP;
    const int r = a.rAgg;
    uint32_t sum = 0;
    if (r == 1) {
        // the reference's window (aggSize = 2, ng_sgm.m:20): the sample column depends on ax only and the sample row on ay
        // only (:417-418), so the 18 double -> int conversions of the 9 taps are 6
        int x2v[3], y2v[3];
        bool xok[3], yokv[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int x1 = x + t - 1, y1 = y + t - 1;
            x2v[t] = f64_to_i32_x86(__dadd_rn((double)(offx + x1), mvx));
            y2v[t] = f64_to_i32_x86(__dadd_rn((double)(offy + y1), mvy));
            xok[t] = x1 >= 0 && x1 <= W - 1 && x2v[t] >= 0 && x2v[t] <= W - 1;
            yokv[t] = y1 >= 0 && y1 <= H - 1 && y2v[t] >= 0 && y2v[t] <= H - 1;
        }
#pragma unroll
        for (int ty = 0; ty < 3; ty++)
#pragma unroll
            for (int tx = 0; tx < 3; tx++) {
                if (yokv[ty] && xok[tx])
                    sum += __popc(cen1[(size_t)W * (y + ty - 1) + (x + tx - 1)] ^ cen2[(size_t)W * y2v[ty] + x2v[tx]]);
                else
                    sum += 5;
            }
    } else {
        for (int ay = -r; ay <= r; ay++) {
            const int y1 = y + ay;
            const int y2 = f64_to_i32_x86(__dadd_rn((double)(offy + y1), mvy));
            const bool yok = y1 >= 0 && y1 <= H - 1 && y2 >= 0 && y2 <= H - 1;
            for (int ax = -r; ax <= r; ax++) {
                const int x1 = x + ax;
                const int x2 = f64_to_i32_x86(__dadd_rn((double)(offx + x1), mvx));
                if (yok && x1 >= 0 && x1 <= W - 1 && x2 >= 0 && x2 <= W - 1)
                    sum += __popc(cen1[(size_t)W * y1 + x1] ^ cen2[(size_t)W * y2 + x2]);
                else
                    sum += 5;
            }
        }
    }
    const int win = (2 * r + 1) * (2 * r + 1);
    Cand o;
    o.cost = f64_to_i32_x86(__dadd_rn(__ddiv_rn(__dmul_rn(1.0, (double)sum), (double)win), 0.5));   // :432
    o.mvx = f64_to_i32_x86(__dadd_rn(mvx, (double)offx));                                           // :433
    o.mvy = f64_to_i32_x86(__dadd_rn(mvy, (double)offy));                                           // :434
    a.C[f * (size_t)NP * D + (size_t)p * D + d] = o;
    // the fast matcher compares motion vectors as packed 16-bit pairs (ng_pack_mv): tell it when one does not fit
    if (a.unsafe && !(o.mvx > -0x3FF0 && o.mvx < 0x3FF0 && o.mvy > -0x3FF0 && o.mvy < 0x3FF0)) atomicOr(a.unsafe, 1u);
}

// The same list for the reference's sizes (3x3 expansion, 3x3 cost window: ng_sgm.m:19-20), one thread = one
// (pixel, hint).  The sample column depends on offx + ax only and the sample row on offy + ay only (:417-418), so
// the 9 candidates of a hint read a 5x5 patch of image-2 census codes: 25 + 9 gathered loads and 10 double -> int
// conversions per 9 candidates instead of 162 and 54 (the per-candidate kernel is bound by its gathered loads).
// The rounded mean of the taps, (int)(1.0 * sum / 9 + 0.5) (:432), is (2 sum + 9) / 18 in integers: 2 sum + 9 is odd,
// so the quotient sum / 9 + 0.5 is never closer than 1/18 to an integer and no rounding of the double division can
// cross one.  The wave's 64 x 27 output dwords go through LDS and leave as contiguous stores.
//
// K4OUT (round 4): a candidate leaves as ONE dword, ng_key4(mvx, mvy, cost) -- 13 + 13 + 5 bits, the key the dedupe kernel used
// to build from the 12-byte entry -- when every motion vector lies inside +-4095 (the launch's flags word 1 is raised otherwise,
// and a second launch of this kernel, the 12-byte form, gated on that flag, writes the Cand list for the general kernels).
// 151 MB per 1242x375 frame written here and read by the dedupe kernel instead of 453: the two kernels together were
// 2.6 of 7.4 ms per batch of 8, most of it those bytes.
__device__ __forceinline__ uint32_t ng_key4(int mvx, int mvy, uint32_t cost) {
    return ((uint32_t)(mvx + 0x1000) << 18) | ((uint32_t)(mvy + 0x1000) << 5) | cost;
}
__device__ __forceinline__ Cand ng_unkey4(uint32_t k) {
    Cand c;
    c.mvx = (int)(k >> 18) - 0x1000; c.mvy = (int)((k >> 5) & 0x1FFFu) - 0x1000; c.cost = (int)(k & 31u);
    return c;
}

template <bool K4OUT>
__global__ __launch_bounds__(256) void ng_cost_hint_kernel(NgCostArgs a) {
    constexpr int OW = K4OUT ? 9 : 27;                                     // dwords a (pixel, hint) leaves
    __shared__ __attribute__((aligned(16))) uint32_t sOut[4][64 * OW];
    if (!K4OUT && a.flags && a.flags[1] == 0) return;                      // the 12-byte list is only needed when a key does not fit
    const int W = a.W, H = a.H;
    const int NP = W * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long total = (long long)NP * 9;
    const long long wbase = ((long long)blockIdx.x * 4 + wave) * 64;       // first (pixel, hint) of this wave
    if (wbase >= total) return;                                            // wave-uniform
    const long long gid = min(wbase + lane, total - 1);                    // lanes past the end redo the last one, nothing of theirs is stored
    const int p = (int)(gid / 9), h = (int)(gid - (long long)p * 9);
    const int y = p / W, x = p - y * W;
    const int dy = (h / 3 - 1) * 8, dx = (h % 3 - 1) * 8;
    const size_t f = blockIdx.y;
    const double* mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
    const double* mvyp = mvxp + (size_t)a.mvW * a.mvH;
    const int yn = clampi(y + dy, 0, a.mvH - 1), xn = clampi(x + dx, 0, a.mvW - 1);
    const double mvx = mvxp[(size_t)a.mvW * yn + xn], mvy = mvyp[(size_t)a.mvW * yn + xn];
    const uint32_t* __restrict__ cen1 = a.cen1 + f * (size_t)NP;
    const uint32_t* __restrict__ cen2 = a.cen2 + f * (size_t)NP;
    int X2[5], Y2[5];
    uint32_t xin = 0, yin = 0, tin = 0;                                    // which sample columns / rows / image-1 taps exist
#pragma unroll
    for (int k = 0; k < 5; k++) {
        X2[k] = f64_to_i32_x86(__dadd_rn((double)(x + k - 2), mvx));       // :418 with offx + ax = k - 2
        Y2[k] = f64_to_i32_x86(__dadd_rn((double)(y + k - 2), mvy));       // :417
        const bool xo = X2[k] >= 0 && X2[k] <= W - 1, yo = Y2[k] >= 0 && Y2[k] <= H - 1;
        xin |= (uint32_t)xo << k; yin |= (uint32_t)yo << k;
        X2[k] = xo ? X2[k] : 0; Y2[k] = yo ? Y2[k] : 0;
    }
    uint32_t T[9], P[25];
#pragma unroll
    for (int ty = 0; ty < 3; ty++)
#pragma unroll
        for (int tx = 0; tx < 3; tx++) {
            const int y1 = y + ty - 1, x1 = x + tx - 1;
            const bool in = y1 >= 0 && y1 <= H - 1 && x1 >= 0 && x1 <= W - 1;
            T[ty * 3 + tx] = cen1[(size_t)W * (in ? y1 : y) + (in ? x1 : x)];
            tin |= (uint32_t)in << (ty * 3 + tx);
        }
#pragma unroll
    for (int ky = 0; ky < 5; ky++)
#pragma unroll
        for (int kx = 0; kx < 5; kx++) P[ky * 5 + kx] = cen2[(size_t)W * Y2[ky] + X2[kx]];
    uint32_t* out = sOut[wave] + lane * OW;
    int cmx[3], cmy[3];
    bool far = false, far4 = false;
#pragma unroll
    for (int o = 0; o < 3; o++) {
        cmx[o] = f64_to_i32_x86(__dadd_rn(mvx, (double)(o - 1)));          // :433
        cmy[o] = f64_to_i32_x86(__dadd_rn(mvy, (double)(o - 1)));          // :434
        far |= !(cmx[o] > -0x3FF0 && cmx[o] < 0x3FF0 && cmy[o] > -0x3FF0 && cmy[o] < 0x3FF0);
        far4 |= !(cmx[o] > -0x1000 && cmx[o] < 0x1000 && cmy[o] > -0x1000 && cmy[o] < 0x1000);
    }
    auto put = [&](const int ox, const int oy, const uint32_t sum) {
        const uint32_t cost = (2u * sum + 9u) / 18u;                       // <= 24 (out-of-image taps count 5)
        if (K4OUT) out[ox * 3 + oy] = ng_key4(cmx[ox], cmy[oy], cost);
        else { uint32_t* o = out + (ox * 3 + oy) * 3; o[0] = (uint32_t)cmx[ox]; o[1] = (uint32_t)cmy[oy]; o[2] = cost; }
    };
    const bool clean = xin == 31u && yin == 31u && tin == 511u;
    if (__builtin_amdgcn_ballot_w64(!clean) == 0) {
#pragma unroll
        for (int ox = 0; ox < 3; ox++)
#pragma unroll
            for (int oy = 0; oy < 3; oy++) {
                uint32_t sum = 0;
#pragma unroll
                for (int ty = 0; ty < 3; ty++)
#pragma unroll
                    for (int tx = 0; tx < 3; tx++) sum += __popc(T[ty * 3 + tx] ^ P[(oy + ty) * 5 + ox + tx]);
                put(ox, oy, sum);
            }
    } else {
#pragma unroll
        for (int ox = 0; ox < 3; ox++)
#pragma unroll
            for (int oy = 0; oy < 3; oy++) {
                uint32_t sum = 0;
#pragma unroll
                for (int ty = 0; ty < 3; ty++)
#pragma unroll
                    for (int tx = 0; tx < 3; tx++) {
                        const uint32_t ok = (tin >> (ty * 3 + tx)) & (xin >> (ox + tx)) & (yin >> (oy + ty)) & 1u;   // :405-421
                        sum += ok ? (uint32_t)__popc(T[ty * 3 + tx] ^ P[(oy + ty) * 5 + ox + tx]) : 5u;
                    }
                put(ox, oy, sum);
            }
    }
    // the fast matcher compares motion vectors as packed 16-bit pairs (ng_pack_mv): tell it when one does not fit
    if (a.unsafe && far) atomicOr(a.unsafe, 1u);
    if (K4OUT && far4 && a.flags[1] == 0) atomicOr(&a.flags[1], 1u);      // a key that does not fit: the launch falls back to 12-byte entries
    __builtin_amdgcn_wave_barrier();
    const int ndw = (int)min((long long)64, total - wbase) * OW;           // dwords this wave owns
    uint32_t* dst = (K4OUT ? a.K4 + f * (size_t)NP * 81 : (uint32_t*)(a.C + f * (size_t)NP * 81)) + (size_t)wbase * OW;
    if (((uintptr_t)dst & 15u) == 0) {                                     // wave-uniform
        for (int i = lane * 4; i < ndw; i += 256) {
            if (i + 4 <= ndw) *(uint4*)(dst + i) = *(const uint4*)(sOut[wave] + i);
            else for (int k = i; k < ndw; k++) dst[k] = sOut[wave][k];
        }
    } else {
        for (int i = lane; i < ndw; i += 64) dst[i] = sOut[wave][i];
    }
}

// One matcher step shared by both variants (calc_pyd_cost_sgm_ng.cpp:39-78 /
// calc_cost_sgm_ng.cpp:46-83): the caller's lanes stride over the current candidates; Lpre is in
// LDS.  Returns the int path cost (C.cost + best - m), not narrowed.
__device__ __forceinline__ int ng_match(const Cand* pre, int D, int mvx, int mvy, int ccost,
                                        uint32_t m, uint32_t jump, int P1) {
    uint32_t min1 = jump, min2 = jump;
    for (int d2 = 0; d2 < D; d2++) {
        const Cand q = pre[d2];                              // same address in every lane: LDS broadcast
        if (mvx == q.mvx && mvy == q.mvy) min1 = (uint32_t)q.cost & 0xFF;            // last match wins
        else if (near2(mvx, q.mvx) && near2(mvy, q.mvy)) min2 = min(min2, (uint32_t)(q.cost + P1) & 0xFF);
    }
    const uint32_t best = min(jump, min(min1, min2));
    return (ccost + (int)best) - (int)m;
}

// =============================================================================================
// aggregation of the hint-map variant (calc_pyd_cost_sgm_ng.cpp:101-278): 2 passes x 2 paths
// (enableDiagnalPath=false :122, adpativeP2=false :120).  One wave per line, S += L atomically
// (u32 adds commute, so the result does not depend on the order the four paths arrive in).
// =============================================================================================
__global__ __launch_bounds__(256) void ng_agg_kernel(NgAggArgs a) {
    __shared__ Cand sL[4][2][FSGM_NG_MAX_D + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int slot = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if ((int)blockIdx.x >= a.blk_begin[i]) slot = i;
    const int base = slot & 1;                               // 0: along x, 1: along y
    const bool mirror = slot >= 2;
    const int W = a.W, H = a.H, D = a.D;
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    const int line = ((int)blockIdx.x - a.blk_begin[slot]) * 4 + wave;
    if (line >= nlines) return;
    const size_t f = blockIdx.y;
    const Cand* __restrict__ Cf = a.C + f * (size_t)NP * D;
    uint32_t* __restrict__ Sf = a.S + f * (size_t)NP * D;
    Cand* pre = sL[wave][0];
    Cand* cur = sL[wave][1];
    uint32_t m = 0;
    for (int t = 0; t < len; t++) {
        int x = base == 0 ? t : line, y = base == 0 ? line : t;
        if (mirror) { x = W - 1 - x; y = H - 1 - y; }
        const size_t off = ((size_t)y * W + x) * D;
        uint32_t lo = 255;
        const uint32_t jump = (m + (uint32_t)a.P2) & 0xFF;
        for (int d = lane; d < D; d += 64) {
            const Cand c = Cf[off + d];
            Cand o = c;
            if (t > 0) {
                o.cost = ng_match(pre, D, c.mvx, c.mvy, c.cost, m, jump, a.P1);
                lo = min(lo, (uint32_t)o.cost & 0xFF);                           // :74 narrowed
            }
            cur[d] = o;
            atomicAdd(&Sf[off + d], (uint32_t)o.cost);                          // :249
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, s));
        m = t > 0 ? lo : 0u;                                                     // :172 / :77
        __builtin_amdgcn_wave_barrier();
        Cand* tmp = pre; pre = cur; cur = tmp;
    }
}

// =============================================================================================
// The same aggregation for D <= 128 (the reference's D = 81), restructured for latency:
//   * the previous pixel's entries sit in LDS as four arrays (mvx, mvy, cost & 0xFF, (cost + P1) & 0xFF)
//     and are read four at a time (ds_read_b128 of one address = broadcast), all reads of a group of
//     four before their compares: the plain loop above waits for every 12-byte read in turn;
//   * "within 2" as one unsigned compare per axis, (mv + 2 - mv') <= 4, valid while no difference can
//     overflow: the cost kernel raises a flag when any |mv| >= 2^30, and then the exact 64-bit form of
//     ng_match is used for the whole launch;
//   * the next pixel's candidates are requested one step ahead.
// Same results as ng_agg_kernel (the order-free u32 sums are atomic adds there and here).
// =============================================================================================
struct NgPre { const int32_t* x; const int32_t* y; const uint32_t* c8; const uint32_t* cp; };

// grid form of the matcher (ng_agg_grid_kernel): a pixel whose motion vectors fit a box of NG_GB x NG_GB stages
// its entries in a grid of that box plus 4 cells around it
constexpr int NG_GB = 12;
constexpr int NG_GS = NG_GB + 8;
constexpr int NG_GCELLS = NG_GS * NG_GS;
constexpr uint32_t NG_BOX_WIDE = 0xFFFFFFFFu;
constexpr uint32_t NG_GRID_MIN_K = 16;   // mean list length from which the grid form is the faster one (it costs the same at any length)

constexpr uint32_t NG_COMPACT_MAX_K = 40;   // mean list length up to which the compact kernel (work ~ K^2) beats the grid form (flat)

// Choice between the aggregation kernels of a launch, made on the device: all candidates are launched, each sums the
// 256 partial list-length sums of the dedupe kernel, reads its flags, and those the lists do not favour return
// (block-uniform).  Compact (one wave a line over the kept entries only): every list <= 64 entries, every entry inside
// the packed key's range, mean length below NG_COMPACT_MAX_K.  Otherwise grid from a mean length of NG_GRID_MIN_K, list
// below it; NG_ROLE_REST: whatever runs when the compact kernel does not (the split kernel of one or two frames).
__device__ __forceinline__ bool ng_agg_not_mine(const NgAggArgs& a, uint32_t* scratch) {
    if (a.role == NG_ROLE_ANY) return false;
    if (threadIdx.x == 0) *scratch = 0;
    __syncthreads();
    uint32_t v = threadIdx.x < 256 ? a.kstat[threadIdx.x] : 0u;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += (uint32_t)__shfl_xor((int)v, s);
    if ((threadIdx.x & 63) == 0) atomicAdd(scratch, v);
    __syncthreads();
    const unsigned long long npix = ((((unsigned long long)a.W * a.H * gridDim.y + 3) / 4 + 15) / 16) * 4;   // the dedupe kernel's sample: every 16th workgroup of 4 pixels
    const unsigned long long sum = *scratch;
    const bool high = sum >= (unsigned long long)NG_GRID_MIN_K * npix;
    const uint32_t flags = a.kstat[256];
    const bool compact = a.with_compact && (flags & 3u) == 0u && sum < (unsigned long long)NG_COMPACT_MAX_K * npix;
    __syncthreads();
    switch (a.role) {
        // one launch per lanes-a-line class: by the mean length for batches (throughput: more lines a wave), 64 for one or two
        // frames (their long lines are serial chains: one line a wave is the shortest step)
        case NG_ROLE_COMPACT: return !compact || (!a.compact_force && a.compact_g != (gridDim.y <= 2 ? 64 : sum < 14ull * npix ? 16 : sum < 28ull * npix ? 32 : 64));
        case NG_ROLE_GRID: return compact || !high;
        case NG_ROLE_LIST: return compact || high;
        default: return compact;                                 // NG_ROLE_REST
    }
}

// motion vector as 2 x u16 (valid for |mv| < 0x3FF0: the launch's unsafe flag is raised otherwise)
constexpr uint32_t NG_PADKEY = 0xC000C000u;   // a staged key no candidate in the packed range is equal or near to
__device__ __forceinline__ uint32_t ng_pack_mv(int mvx, int mvy) {
    return ((uint32_t)(mvx + 0x4000) << 16) | ((uint32_t)(mvy + 0x4000) & 0xFFFFu);
}

__device__ __forceinline__ int ng_match4(const NgPre& q, int D, int mvx, int mvy, int ccost, uint32_t m, uint32_t jump, bool safe) {
    uint32_t min1 = jump, min2 = jump;
    if (safe) {
        // q.x holds packed keys (ng_pack_mv): t = c + (2,2) - q per 16-bit half; equal is t == (2,2), within 2 on
        // both axes is both halves <= 4.  8 VALU per entry, no scalar mask arithmetic.
        const uint32_t ck2 = ng_pack_mv(mvx, mvy) + 0x00020002u;
        const uint32_t* qk = (const uint32_t*)q.x;
        uint32_t near2min = 0xFFFFu;
        int d2 = 0;
        for (; d2 + 4 <= D; d2 += 4) {
            const uint4 k4 = *(const uint4*)(qk + d2);
            const uint4 c8 = *(const uint4*)(q.c8 + d2), cp = *(const uint4*)(q.cp + d2);
            const uint32_t ka[4] = {k4.x, k4.y, k4.z, k4.w};
            const uint32_t c8a[4] = {c8.x, c8.y, c8.z, c8.w}, cpa[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t t = pk_sub(ck2, ka[i]);
                const bool nr = pk_min(t, 0x00040004u) == t, eq = t == 0x00020002u;
                min1 = eq ? c8a[i] : min1;                                        // last match wins
                uint32_t sel = nr ? cpa[i] : 0xFFFFu;
                sel = eq ? 0xFFFFu : sel;
                near2min = min(near2min, sel);
            }
        }
        for (; d2 < D; d2++) {
            const uint32_t t = pk_sub(ck2, qk[d2]);
            const bool nr = pk_min(t, 0x00040004u) == t, eq = t == 0x00020002u;
            min1 = eq ? q.c8[d2] : min1;
            if (nr && !eq) near2min = min(near2min, q.cp[d2]);
        }
        min2 = min(min2, near2min);
    } else {
        for (int d2 = 0; d2 < D; d2++) {
            const int qx = q.x[d2], qy = q.y[d2];
            if (mvx == qx && mvy == qy) min1 = q.c8[d2];
            else if (near2(mvx, qx) && near2(mvy, qy)) min2 = min(min2, q.cp[d2]);
        }
    }
    const uint32_t best = min(jump, min(min1, min2));
    return (ccost + (int)best) - (int)m;
}

// Two candidates of one lane against the same predecessor (D % 4 == 0, motion vectors in the safe range):
// one pass over the entries, every 16-byte LDS read shared, the loop unrolled so the reads run ahead.
__device__ __forceinline__ void ng_match4_pair(const NgPre& q, int D, int mvxa, int mvya, int mvxb, int mvyb, uint32_t jump,
                                               uint32_t& besta, uint32_t& bestb) {
    uint32_t min1a = jump, min2a = jump, min1b = jump, min2b = jump;
    const uint32_t axa = (uint32_t)mvxa + 2u, aya = (uint32_t)mvya + 2u, axb = (uint32_t)mvxb + 2u, ayb = (uint32_t)mvyb + 2u;
#pragma unroll 9
    for (int d2 = 0; d2 < D; d2 += 4) {
        const int4 qx = *(const int4*)(q.x + d2), qy = *(const int4*)(q.y + d2);
        const uint4 c8 = *(const uint4*)(q.c8 + d2), cp = *(const uint4*)(q.cp + d2);
        const int qxa[4] = {qx.x, qx.y, qx.z, qx.w}, qya[4] = {qy.x, qy.y, qy.z, qy.w};
        const uint32_t c8a[4] = {c8.x, c8.y, c8.z, c8.w}, cpa[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool eqa = mvxa == qxa[i] && mvya == qya[i];
            const bool nra = (axa - (uint32_t)qxa[i]) <= 4u && (aya - (uint32_t)qya[i]) <= 4u;
            min1a = eqa ? c8a[i] : min1a;                                     // last match wins
            min2a = (nra && !eqa) ? min(min2a, cpa[i]) : min2a;
            const bool eqb = mvxb == qxa[i] && mvyb == qya[i];
            const bool nrb = (axb - (uint32_t)qxa[i]) <= 4u && (ayb - (uint32_t)qya[i]) <= 4u;
            min1b = eqb ? c8a[i] : min1b;
            min2b = (nrb && !eqb) ? min(min2b, cpa[i]) : min2b;
        }
    }
    besta = min(jump, min(min1a, min2a));
    bestb = min(jump, min(min1b, min2b));
}

// Repeats in a pixel's candidate list.  The 9 hints of a pixel are samples of a smooth map 8 pixels apart, so
// their 3x3 expansions overlap: in real flow fields most of the 81 candidates are repeats of one another.
// Two candidates with the same motion vector and the same cost C get the same path cost on every path (the
// matcher's result depends on the motion vector alone: calc_pyd_cost_sgm_ng.cpp:39-78), so a predecessor list
// without repeats gives every candidate the same minima as the full list, as long as of every group of repeats
// the LAST is kept ("last exact match wins", :60-62: the kept entries stay in their order).  One wave per pixel:
// keys (mvx, mvy, C) packed into 31 bits, every lane finds the last index holding its key, kept entries are
// ranked with two ballots.  A pixel with a vector outside +-4095 keeps its whole list.
//
// "The last index holding my key" without comparing all pairs: up to three rounds over a 256-slot table in LDS.
// In a round every unsettled entry posts its index + 1 to the slot its key hashes to (ds_max), then reads the slot's
// winner and that entry's key.  Entries with one key share a slot, so either the winner carries their key -- it is
// then the last of them and the whole group is settled -- or it belongs to another key and the whole group goes on
// to the next round with another hash.  81 keys in 256 slots leave a few groups for round two and next to none
// for round three; what is left after that takes the all-pairs scan (wave-uniform branch).
//
// Also written per pixel: the bounding box of its motion vectors, as its packed origin (ng_pack_mv) when both
// sides are <= NG_GB, NG_BOX_WIDE otherwise -- what the grid form of the matcher (ng_agg_grid_kernel) needs.
__global__ __launch_bounds__(256) void ng_dedupe_kernel(const Cand* __restrict__ C, const uint32_t* __restrict__ K4, const uint32_t* __restrict__ flags,
                                                        uint16_t* __restrict__ dd, uint8_t* __restrict__ dk,
                                                        uint32_t* __restrict__ dbox, uint32_t* __restrict__ kstat, uint32_t* __restrict__ ck,
                                                        uint16_t* __restrict__ cm, int NPtot, int D) {
    __shared__ __attribute__((aligned(16))) uint32_t sk[4][128];
    __shared__ __attribute__((aligned(16))) uint32_t stab[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wave;
    if (p >= NPtot) return;                                   // wave-uniform
    const Cand* c = C + (size_t)p * D;
    const int d0 = lane, d1 = lane + 64;
    const bool has0 = d0 < D, has1 = d1 < D;
    auto key_of = [](const Cand& e, bool& ok) -> uint32_t {
        ok = e.mvx > -0x1000 && e.mvx < 0x1000 && e.mvy > -0x1000 && e.mvy < 0x1000 && e.cost >= 0 && e.cost < 32;
        return ((uint32_t)(e.mvx + 0x1000) << 18) | ((uint32_t)(e.mvy + 0x1000) << 5) | (uint32_t)e.cost;
    };
    bool ok0 = true, ok1 = true;
    const Cand z = {0, 0, 0};
    Cand e0 = z, e1 = z;
    uint32_t k0, k1;
    if (K4 && flags[1] == 0) {                               // launch-uniform: the cost kernel left the keys themselves (4 bytes an entry)
        k0 = has0 ? K4[(size_t)p * D + d0] : ng_key4(0, 0, 0);
        k1 = has1 ? K4[(size_t)p * D + d1] : ng_key4(0, 0, 0);
        e0 = ng_unkey4(k0); e1 = ng_unkey4(k1);
    } else {
        if (has0) e0 = c[d0];
        if (has1) e1 = c[d1];
        k0 = key_of(e0, ok0); k1 = key_of(e1, ok1);
    }
    sk[wave][d0] = has0 ? k0 : 0xFFFFFFFFu;
    sk[wave][d1] = has1 ? k1 : 0xFFFFFFFFu;
    const bool all_ok = __builtin_amdgcn_ballot_w64(!(ok0 && ok1)) == 0;
    __builtin_amdgcn_wave_barrier();
    if (dbox && !(K4 && flags[1] == 0)) {                     // (4-byte entries: ng_dbox_kernel makes the boxes, and only when a matcher needs them)
        uint32_t box = NG_BOX_WIDE;
        if (all_ok) {                                                      // every |mv| < 4096: biased values are small and positive
            const Cand f0 = has0 ? e0 : c[0], f1 = has1 ? e1 : f0;         // lanes without an entry repeat one that exists
            const uint32_t xl = wave_min_u32((uint32_t)(min(f0.mvx, f1.mvx) + 0x4000)), xh = 0x8000u - wave_min_u32((uint32_t)(0x4000 - max(f0.mvx, f1.mvx)));
            const uint32_t yl = wave_min_u32((uint32_t)(min(f0.mvy, f1.mvy) + 0x4000)), yh = 0x8000u - wave_min_u32((uint32_t)(0x4000 - max(f0.mvy, f1.mvy)));
            if (xh - xl < (uint32_t)NG_GB && yh - yl < (uint32_t)NG_GB) box = (xl << 16) | yl;
        }
        if (lane == 0) dbox[p] = box;
    }
    int last0 = d0, last1 = d1;
    if (all_ok) {
        bool open0 = has0, open1 = has1;
        uint32_t* tab = stab[wave];
#pragma unroll 1
        for (int round = 0; round < 3; round++) {
            if (__builtin_amdgcn_ballot_w64(open0 || open1) == 0) break;
            *(uint4*)(tab + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
            __builtin_amdgcn_wave_barrier();
            const uint32_t mul = round == 0 ? 0x9E3779B1u : round == 1 ? 0x85EBCA77u : 0xC2B2AE3Du;
            const uint32_t s0 = (k0 * mul) >> 24, s1 = (k1 * mul) >> 24;
            if (open0) atomicMax(&tab[s0], (uint32_t)d0 + 1u);
            if (open1) atomicMax(&tab[s1], (uint32_t)d1 + 1u);
            __builtin_amdgcn_wave_barrier();
            if (open0) { const int w = (int)tab[s0] - 1; if (sk[wave][w] == k0) { last0 = w; open0 = false; } }
            if (open1) { const int w = (int)tab[s1] - 1; if (sk[wave][w] == k1) { last1 = w; open1 = false; } }
            __builtin_amdgcn_wave_barrier();
        }
        if (__builtin_amdgcn_ballot_w64(open0 || open1) != 0) {
            last0 = d0; last1 = d1;
            for (int e = 0; e < D; e += 4) {
                const uint4 k4 = *(const uint4*)(&sk[wave][e]);
                const uint32_t ka[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    last0 = ka[i] == k0 ? e + i : last0;
                    last1 = ka[i] == k1 ? e + i : last1;
                }
            }
        }
    }
    const bool keep0 = has0 && last0 == d0, keep1 = has1 && last1 == d1;
    const unsigned long long b0 = __builtin_amdgcn_ballot_w64(keep0), b1 = __builtin_amdgcn_ballot_w64(keep1);
    const int n0 = __popcll(b0);
    // place of entry e among the kept ones; a repeat records the place of the entry it repeats (its last twin, which is kept)
    auto place_of = [&](int e) -> uint32_t {
        return e < 64 ? (uint32_t)__popcll(b0 & ((1ull << e) - 1ull)) : (uint32_t)n0 + (uint32_t)__popcll(b1 & ((1ull << (e - 64)) - 1ull));
    };
    const uint32_t pl0 = place_of(keep0 ? d0 : last0), pl1 = place_of(keep1 ? d1 : last1);
    if (has0) dd[(size_t)p * D + d0] = (uint16_t)(keep0 ? pl0 : 0x8000u | pl0);
    if (has1) dd[(size_t)p * D + d1] = (uint16_t)(keep1 ? pl1 : 0x8000u | pl1);
    const int K = n0 + __popcll(b1);
    if (ck) {
        // The kept entries in place order.  Each stands for its group of repeats; the group's sums live at its FIRST
        // member's index (the WTA takes the first minimum over d, :281-299, and all members of a group tie): first
        // index = minimum over the members, collected per group under the index of its last member.
        uint32_t* fi = stab[wave];                               // (the hash table is done with)
        __builtin_amdgcn_wave_barrier();
        fi[d0] = 0xFFFFFFFFu; fi[d1] = 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
        if (has0) atomicMin(&fi[last0], (uint32_t)d0);
        if (has1) atomicMin(&fi[last1], (uint32_t)d1);
        __builtin_amdgcn_wave_barrier();
        if (keep0) { ck[(size_t)p * D + pl0] = ng_pack_mv(e0.mvx, e0.mvy); cm[(size_t)p * D + pl0] = (uint16_t)((fi[d0] << 8) | ((uint32_t)e0.cost & 0xFFu)); }
        if (keep1) { ck[(size_t)p * D + pl1] = ng_pack_mv(e1.mvx, e1.mvy); cm[(size_t)p * D + pl1] = (uint16_t)((fi[d1] << 8) | ((uint32_t)e1.cost & 0xFFu)); }
    }
    if (lane == 0) {
        dk[p] = (uint8_t)K;
        if (kstat) {
            if ((blockIdx.x & 15) == 0) atomicAdd(&kstat[(blockIdx.x >> 4) & 255], (uint32_t)K);   // a 1-in-16 sample of the pixels
            // list-length classes of the launch (bit 0: a list beyond 64 entries, 2: beyond 16, 3: beyond 32) and bit 1: an entry
            // outside the packed key's range; a bit is written by the first few pixels that find it clear
            const uint32_t bits = (K > 64 ? 1u : 0u) | (all_ok ? 0u : 2u) | (K > 16 ? 4u : 0u) | (K > 32 ? 8u : 0u);
            if (bits & ~__hip_atomic_load(&kstat[256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&kstat[256], bits);
        }
    }
}

// One THREAD per (line, candidate): a 256-thread workgroup advances 256/D lines (3 at D = 81, 95 % of the
// lanes busy; one wave per line leaves 37 % idle and needs two rounds).  The per-line minimum crosses
// waves through an LDS atomicMin in a three-slot ring (written at step t, read at t+1, reset at t+2),
// one barrier per step.
__global__ __launch_bounds__(256) void ng_agg_lines_kernel(NgAggArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sNg[];   // [line][2 buffers][4 arrays][Dp], then [line][8]: 3 minima, the staged lists' lengths at [4 + step parity]
    if (ng_agg_not_mine(a, sNg)) return;
    int slot = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if ((int)blockIdx.x >= a.blk_begin[i]) slot = i;
    const int base = slot & 1;                               // 0: along x, 1: along y
    const bool mirror = slot >= 2;
    const int W = a.W, H = a.H, D = a.D, Dp = (D + 3) & ~3;
    const int LPB = 256 / D;                                  // lines per workgroup
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    const int g = threadIdx.x;
    const int ll = min(g / D, LPB - 1), cand = g - (g / D) * D;
    const bool tact = g < LPB * D;                            // this thread holds a candidate
    const int line = ((int)blockIdx.x - a.blk_begin[slot]) * LPB + ll;
    const bool lact = tact && line < nlines;
    const int linec = min(line, nlines - 1);
    const size_t f = blockIdx.y;
    const Cand* __restrict__ Cf = a.C + f * (size_t)NP * D;
    uint32_t* __restrict__ Sf = a.S + f * (size_t)NP * D;
    uint32_t* buf0 = sNg + (size_t)ll * 8 * Dp;
    uint32_t* buf1 = buf0 + 4 * Dp;
    uint32_t* smin = sNg + (size_t)LPB * 8 * Dp + ll * 8;     // [0..2] minima ring, [4 + (t & 1)] entries staged by step t
    if (cand < 3 && tact) smin[cand] = 255u;
    const bool safe = *a.unsafe == 0;                         // no motion vector of this launch near the int range
    // the line as a walk over entry indices (pixel * D + candidate < 2^31, checked by the host entry points): step t
    // sits at e_cur, the fetch cursor PF steps ahead (it stops at the last pixel)
    int pix0 = base == 0 ? linec * W : linec;
    if (mirror) pix0 = NP - 1 - pix0;
    const int dpix = (mirror ? -1 : 1) * (base == 0 ? 1 : W);
    const uint32_t dent = (uint32_t)(dpix * D);              // mod 2^32: the indices themselves stay in range
    uint32_t e_cur = (uint32_t)(pix0 * D + cand), p_fet = (uint32_t)pix0, e_fet = e_cur;
    const uint16_t* __restrict__ ddf = a.dd ? a.dd + f * (size_t)NP * D : nullptr;
    const uint8_t* __restrict__ dkf = a.dk ? a.dk + f * (size_t)NP : nullptr;
    // the candidates of step t+PF are requested while step t computes (a gathered 12-byte load takes
    // longer than one step); the ring rotates by unrolling, not by copying (a copy would wait for the load)
    constexpr int PF = 4;
    Cand ring[PF];
    uint32_t rpl[PF], rlen[PF];                               // the candidate's place in its pixel's list without repeats, that list's length
#pragma unroll
    for (int k = 0; k < PF; k++) {
        ring[k] = Cf[e_fet];
        rpl[k] = ddf ? ddf[e_fet] : (uint32_t)cand;
        rlen[k] = dkf ? dkf[p_fet] : (uint32_t)D;
        if (k + 1 < len) { p_fet += (uint32_t)dpix; e_fet += dent; }
    }
    __syncthreads();
    // one step with its ring slot named by the caller: the steady-state loop is straight-line code (see ng_agg_compact_kernel)
    auto step = [&](const int t, Cand& rg, uint32_t& rp, uint32_t& rl) {
        const Cand c = rg;
        const uint32_t place = rp, K = rl;
        rg = Cf[e_fet];
        rp = ddf ? ddf[e_fet] : (uint32_t)cand;
        rl = dkf ? dkf[p_fet] : (uint32_t)D;
        if (t + PF + 1 < len) { p_fet += (uint32_t)dpix; e_fet += dent; }
        const NgPre q{(const int32_t*)buf0, (const int32_t*)buf0 + Dp, buf0 + 2 * Dp, buf0 + 3 * Dp};
        const uint32_t m = t >= 2 ? smin[(t - 1) % 3] : 0u;    // :172 / :77; stored minimum 0 at a path start
        const uint32_t jump = (m + (uint32_t)a.P2) & 0xFF;
        int o = c.cost;
        if (t > 0) {
            const int Kpre = (int)smin[4 + ((t - 1) & 1)];
            o = ng_match4(q, safe ? (Kpre + 3) & ~3 : Kpre, c.mvx, c.mvy, c.cost, m, jump, safe);
            if (tact) atomicMin(&smin[t % 3], (uint32_t)o & 0xFF);                // :74 narrowed
        }
        if (tact) {
            if (place < 0x8000u) {                                    // a kept entry (repeats carry 0x8000 | the place of the entry they repeat)
                buf1[place] = safe ? ng_pack_mv(c.mvx, c.mvy) : (uint32_t)c.mvx; buf1[Dp + place] = (uint32_t)c.mvy;
                buf1[2 * Dp + place] = (uint32_t)o & 0xFF; buf1[3 * Dp + place] = (uint32_t)(o + a.P1) & 0xFF;
            }
            if (cand < 3 && K + cand < (uint32_t)Dp) {         // neutral entries up to the next multiple of 4
                buf1[K + cand] = NG_PADKEY; buf1[Dp + K + cand] = 0x7FFFFFFFu;
                buf1[2 * Dp + K + cand] = 0xFFFFu; buf1[3 * Dp + K + cand] = 0xFFFFu;
            }
            if (lact) atomicAdd(&Sf[e_cur], (uint32_t)o);                         // :249
            if (cand == 0) { smin[(t + 1) % 3] = 255u; smin[4 + (t & 1)] = K; }
        }
        e_cur += dent;
        __syncthreads();
        uint32_t* tmp = buf0; buf0 = buf1; buf1 = tmp;
    };
    int t0 = 0;
    for (; t0 + PF <= len; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) step(t0 + u, ring[u], rpl[u], rlen[u]);
    }
#pragma unroll
    for (int u = 0; u < PF - 1; u++)
        if (t0 + u < len) step(t0 + u, ring[u], rpl[u], rlen[u]);                  // block-uniform
}

// ng_agg_lines_kernel with a second form of the matcher.  What a candidate needs from the predecessor's list is the
// cost of the last entry with its own motion vector and the smallest cost + P1 among the entries within 2 on both
// axes (:39-78) -- a 5x5 neighbourhood in motion-vector space.  A pixel whose motion vectors fit a NG_GB x NG_GB box
// (launch_ng_dedupe writes the box; smooth hint maps always do) also stages its kept entries in a grid over that box
// plus 4 cells around it: per cell the smallest cost + P1 (ds_min) and (place + 1) << 8 | cost of the last entry
// (ds_max, places grow with the candidate index).  A candidate of the next pixel then reads 25 cells instead of
// walking the list -- ~60 instructions against 8 per entry -- and one farther than 2 from the box matches nothing.
// Three grids per line rotate (read the one staged a step ago, stage, clear the third), so a step still has one
// barrier.  The lists are staged as before, and a step whose predecessors (of any of the workgroup's lines) did not
// fit a box takes the list matcher: same results either way.
__global__ __launch_bounds__(256) void ng_agg_grid_kernel(NgAggArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sNg[];   // lists [line][2][4][Dp]; grids [line][3][2][NG_GCELLS]; [line][8] minima ring + lengths; [2][line] box flags
    if (ng_agg_not_mine(a, sNg)) return;
    int slot = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if ((int)blockIdx.x >= a.blk_begin[i]) slot = i;
    slot = a.slot_of[slot];
    const int base = slot & 1;                               // 0: along x, 1: along y
    const bool mirror = slot >= 2;
    const int W = a.W, H = a.H, D = a.D, Dp = (D + 3) & ~3;
    const int LPB = 256 / D;                                  // lines per workgroup
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    const int g = threadIdx.x;
    const int ll = min(g / D, LPB - 1), cand = g - (g / D) * D;
    const bool tact = g < LPB * D;                            // this thread holds a candidate
    int bb = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if ((int)blockIdx.x >= a.blk_begin[i]) bb = i;
    const int line = ((int)blockIdx.x - a.blk_begin[bb]) * LPB + ll;
    const bool lact = tact && line < nlines;
    const int linec = min(line, nlines - 1);
    const size_t f = blockIdx.y;
    const Cand* __restrict__ Cf = a.C + f * (size_t)NP * D;
    uint32_t* __restrict__ Sf = a.S + f * (size_t)NP * D;
    uint32_t* buf0 = sNg + (size_t)ll * 8 * Dp;
    uint32_t* buf1 = buf0 + 4 * Dp;
    uint32_t* const grids = sNg + (size_t)LPB * 8 * Dp + (size_t)ll * 6 * NG_GCELLS;
    uint32_t* gpre = grids;                                   // staged by the previous step: [0..CELLS) smallest cost + P1, [CELLS..2 CELLS) last entry
    uint32_t* gcur = grids + 2 * NG_GCELLS;
    uint32_t* gnxt = grids + 4 * NG_GCELLS;
    uint32_t* const tailw = sNg + (size_t)LPB * 8 * Dp + (size_t)LPB * 6 * NG_GCELLS;
    uint32_t* smin = tailw + ll * 8;                          // [0..2] minima ring, [4 + (t & 1)] entries staged by step t
    uint32_t* const sflag = tailw + LPB * 8;                  // [t & 1][line]: the pixel of step t fits a box
    if (cand < 3 && tact) smin[cand] = 255u;
    if (tact)
        for (int i = cand; i < 3 * NG_GCELLS / 2; i += D) {   // uint2 units; near halves all ones, last halves zero
            const int b = i / (NG_GCELLS / 2), r = i - b * (NG_GCELLS / 2);
            *(uint2*)(grids + b * 2 * NG_GCELLS + 2 * r) = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            *(uint2*)(grids + b * 2 * NG_GCELLS + NG_GCELLS + 2 * r) = make_uint2(0u, 0u);
        }
    const bool safe = *a.unsafe == 0;                         // no motion vector of this launch near the int range
    // the line as a walk over entry indices (pixel * D + candidate < 2^31, checked by the host entry points): step t
    // sits at e_cur, the fetch cursor PF steps ahead (it stops at the last pixel)
    int pix0 = base == 0 ? linec * W : linec;
    if (mirror) pix0 = NP - 1 - pix0;
    const int dpix = (mirror ? -1 : 1) * (base == 0 ? 1 : W);
    const uint32_t dent = (uint32_t)(dpix * D);              // mod 2^32: the indices themselves stay in range
    uint32_t e_cur = (uint32_t)(pix0 * D + cand), p_fet = (uint32_t)pix0, e_fet = e_cur;
    const uint16_t* __restrict__ ddf = a.dd + f * (size_t)NP * D;
    const uint8_t* __restrict__ dkf = a.dk + f * (size_t)NP;
    const uint32_t* __restrict__ dbf = a.dbox + f * (size_t)NP;
    constexpr int PF = 4;
    Cand ring[PF];
    uint32_t rpl[PF], rlen[PF], rbox[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        ring[k] = Cf[e_fet];
        rpl[k] = ddf[e_fet];
        rlen[k] = dkf[p_fet];
        rbox[k] = dbf[p_fet];
        if (k + 1 < len) { p_fet += (uint32_t)dpix; e_fet += dent; }
    }
    uint32_t pbox = NG_BOX_WIDE;                              // box of the previous step's pixel
    __syncthreads();
    // one step with its ring slot named by the caller: the steady-state loop is straight-line code (see ng_agg_compact_kernel)
    auto step = [&](const int t, Cand& rg, uint32_t& rp, uint32_t& rl, uint32_t& rb) {
        const Cand c = rg;
        const uint32_t place = rp, K = rl, box = rb;
        rg = Cf[e_fet];
        rp = ddf[e_fet];
        rl = dkf[p_fet];
        rb = dbf[p_fet];
        if (t + PF + 1 < len) { p_fet += (uint32_t)dpix; e_fet += dent; }
        const uint32_t m = t >= 2 ? smin[(t - 1) % 3] : 0u;    // :172 / :77; stored minimum 0 at a path start
        const uint32_t jump = (m + (uint32_t)a.P2) & 0xFF;
        int o = c.cost;
        if (t > 0) {
            bool grid = true;
            for (int i = 0; i < LPB; i++) grid = grid && sflag[((t - 1) & 1) * LPB + i] != 0;   // block-uniform
            if (grid) {
                // cell of this candidate in the predecessor's grid (mod 2^32: exact, the box origin is small)
                const uint32_t gx = (uint32_t)c.mvx + 0x4000u - (pbox >> 16) + 4u, gy = (uint32_t)c.mvy + 0x4000u - (pbox & 0xFFFFu) + 4u;
                uint32_t best = jump;
                if (gx - 2u < (uint32_t)(NG_GS - 4) && gy - 2u < (uint32_t)(NG_GS - 4)) {
                    const uint32_t* q = gpre + (gy - 2u) * NG_GS + (gx - 2u);
                    uint32_t nr = 0xFFFFFFFFu;
#pragma unroll
                    for (int r = 0; r < 5; r++)
#pragma unroll
                        for (int k = 0; k < 5; k++)
                            if (r != 2 || k != 2) nr = min(nr, q[r * NG_GS + k]);
                    const uint32_t last = q[NG_GCELLS + 2 * NG_GS + 2];
                    best = min(best, nr);
                    if (last) best = min(best, last & 0xFFu);                    // (a later entry replaces an earlier one: :60-62)
                }
                o = (c.cost + (int)best) - (int)m;
            } else {
                const NgPre q{(const int32_t*)buf0, (const int32_t*)buf0 + Dp, buf0 + 2 * Dp, buf0 + 3 * Dp};
                const int Kpre = (int)smin[4 + ((t - 1) & 1)];
                o = ng_match4(q, safe ? (Kpre + 3) & ~3 : Kpre, c.mvx, c.mvy, c.cost, m, jump, safe);
            }
            if (tact) atomicMin(&smin[t % 3], (uint32_t)o & 0xFF);                // :74 narrowed
        }
        if (tact) {
            if (place < 0x8000u) {                                    // a kept entry (repeats carry 0x8000 | the place of the entry they repeat)
                const uint32_t c8 = (uint32_t)o & 0xFF, cp = (uint32_t)(o + a.P1) & 0xFF;
                buf1[place] = safe ? ng_pack_mv(c.mvx, c.mvy) : (uint32_t)c.mvx; buf1[Dp + place] = (uint32_t)c.mvy;
                buf1[2 * Dp + place] = c8; buf1[3 * Dp + place] = cp;
                if (box != NG_BOX_WIDE) {
                    const uint32_t cell = ((uint32_t)c.mvy + 0x4000u - (box & 0xFFFFu) + 4u) * NG_GS + ((uint32_t)c.mvx + 0x4000u - (box >> 16) + 4u);
                    atomicMin(&gcur[cell], cp);
                    atomicMax(&gcur[NG_GCELLS + cell], ((place + 1u) << 8) | c8);
                }
            }
            if (cand < 3 && K + cand < (uint32_t)Dp) {         // neutral entries up to the next multiple of 4
                buf1[K + cand] = NG_PADKEY; buf1[Dp + K + cand] = 0x7FFFFFFFu;
                buf1[2 * Dp + K + cand] = 0xFFFFu; buf1[3 * Dp + K + cand] = 0xFFFFu;
            }
            if (lact) atomicAdd(&Sf[e_cur], (uint32_t)o);                         // :249
            if (cand == 0) { smin[(t + 1) % 3] = 255u; smin[4 + (t & 1)] = K; sflag[(t & 1) * LPB + ll] = box != NG_BOX_WIDE; }
            for (int i = cand; i < NG_GCELLS / 2; i += D) {                       // the grid staged two steps ago is free
                *(uint2*)(gnxt + 2 * i) = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                *(uint2*)(gnxt + NG_GCELLS + 2 * i) = make_uint2(0u, 0u);
            }
        }
        pbox = box;
        e_cur += dent;
        __syncthreads();
        uint32_t* tmp = buf0; buf0 = buf1; buf1 = tmp;
        tmp = gpre; gpre = gcur; gcur = gnxt; gnxt = tmp;
    };
    int t0 = 0;
    for (; t0 + PF <= len; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) step(t0 + u, ring[u], rpl[u], rlen[u], rbox[u]);
    }
#pragma unroll
    for (int u = 0; u < PF - 1; u++)
        if (t0 + u < len) step(t0 + u, ring[u], rpl[u], rlen[u], rbox[u]);                  // block-uniform
}

// The same aggregation over the KEPT entries only: one wave per line, lane = place in the pixel's list without
// repeats (launch_ng_dedupe writes that list as packed motion vector + (candidate index, cost): ck / cm).  A repeat gets
// the same path cost as the entry it repeats on every path (ng_dedupe_kernel's header), so nothing is lost by not
// computing it: S is added to at the kept entries only and the WTA reads a repeat's sum from the entry it repeats
// (ng_wta_kernel; launch_ng_fill_repeats when S itself is wanted).  With the lists of real hint maps (10-20 of 81
// entries distinct) that is a quarter of the matcher work and of the atomic adds; the line's minimum is a wave
// reduction and a step has no workgroup barrier at all -- the four waves of a workgroup walk four lines independently.
// Runs when every list of the launch has at most 64 entries inside the packed key's range (ng_agg_not_mine).
// four entries of the predecessor against one candidate key: first-class "last exact match wins" (min1 follows the entry
// order) and the minimum over the near, not equal entries
__device__ __forceinline__ void ng_match4_group(const uint4 k4, const uint4 c8, const uint4 cp, const uint32_t ck2, uint32_t& min1, uint32_t& near2min) {
    const uint32_t ka[4] = {k4.x, k4.y, k4.z, k4.w};
    const uint32_t c8a[4] = {c8.x, c8.y, c8.z, c8.w}, cpa[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t t = pk_sub(ck2, ka[i]);
        const bool nr = pk_min(t, 0x00040004u) == t, eq = t == 0x00020002u;
        min1 = eq ? c8a[i] : min1;                                            // last match wins
        uint32_t sel = nr ? cpa[i] : 0xFFFFu;
        sel = eq ? 0xFFFFu : sel;
        near2min = min(near2min, sel);
    }
}

// The reads of a group of four are requested one group ahead of their compares (round 4: with the reads issued and waited
// for inside one iteration a wave parked in s_waitcnt for the LDS round trip once per four entries -- 76 % of its time in the
// counters of round 3)
__device__ __forceinline__ uint32_t ng_match4_key(const uint32_t* qk, const uint32_t* qc8, const uint32_t* qcp, int K4, uint32_t key, uint32_t jump) {
    const uint32_t ck2 = key + 0x00020002u;
    uint32_t min1 = jump, near2min = 0xFFFFu;
    if (K4 <= 0) return jump;
    uint4 k4 = *(const uint4*)qk, c8 = *(const uint4*)qc8, cp = *(const uint4*)qcp;
    for (int d2 = 4; d2 < K4; d2 += 4) {
        const uint4 nk = *(const uint4*)(qk + d2), nc8 = *(const uint4*)(qc8 + d2), ncp = *(const uint4*)(qcp + d2);
        ng_match4_group(k4, c8, cp, ck2, min1, near2min);
        k4 = nk; c8 = nc8; cp = ncp;
    }
    ng_match4_group(k4, c8, cp, ck2, min1, near2min);
    return min(jump, min(min1, near2min));
}

// One line a wave, lists of at most 16 entries on both sides of the step: the four rows of the wave split the predecessor's
// entries between them (row r takes the groups r, r + 4, ... of four), every row working for the same 16 candidates (lane & 15),
// and the rows' results meet through two row swaps.  "Last exact match wins" survives the split as a maximum over
// (place + 1) << 8 | cost.  A lone wave issues one instruction at a time (~6 cycles each): what counts for the
// 1242-step lines of a single frame is the instruction count of a step, 4 x 33 for the groups of a 14-entry list before.
__device__ __forceinline__ uint32_t ng_match4_key_rows(const uint32_t* qk, const uint32_t* qc8, const uint32_t* qcp, int K4, uint32_t key, uint32_t jump, int row) {
    const uint32_t ck2 = key + 0x00020002u;
    uint32_t last = 0, near2min = 0xFFFFu;                                    // last: ((place + 1) << 8) | cost of the last exact match, 0 = none
    const int d2 = 4 * row;
    if (d2 < K4) {                                                            // K4 <= 16: at most one group a row
        const uint4 k4 = *(const uint4*)(qk + d2), c8 = *(const uint4*)(qc8 + d2), cp = *(const uint4*)(qcp + d2);
        const uint32_t ka[4] = {k4.x, k4.y, k4.z, k4.w};
        const uint32_t c8a[4] = {c8.x, c8.y, c8.z, c8.w}, cpa[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t t = pk_sub(ck2, ka[i]);
            const bool nr = pk_min(t, 0x00040004u) == t, eq = t == 0x00020002u;
            last = eq ? (((uint32_t)(d2 + i + 1) << 8) | c8a[i]) : last;      // places ascend inside a row's walk: the later one replaces
            uint32_t sel = nr ? cpa[i] : 0xFFFFu;
            sel = eq ? 0xFFFFu : sel;
            near2min = min(near2min, sel);
        }
    }
    // rows 0 <-> 1, 2 <-> 3, then halves: every lane ends with the results over all entries
    uint32_t l2 = last, n2 = near2min;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(last), "+v"(l2));
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(near2min), "+v"(n2));
    last = max(last, l2); near2min = min(near2min, n2);
    l2 = last; n2 = near2min;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(last), "+v"(l2));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(near2min), "+v"(n2));
    last = max(last, l2); near2min = min(near2min, n2);
    const uint32_t min1 = last ? (last & 0xFFu) : jump;
    return min(jump, min(min1, near2min));
}

// G lanes a line (64 / G lines a wave), G = 16, 32 or 64 by the launch's MEAN list length: a lane takes the places
// pl, pl + G, pl + 2G ... of its line's list, so a list longer than G costs the wave's lines extra rounds at that pixel only
// (lists hold up to 64 entries whatever G).  Every round writes all G slots it covers (entries, then neutral ones), so
// the matcher of a wave can run to the longest of its lines' staged lists.
template <int G>
__global__ __launch_bounds__(256) void ng_agg_compact_kernel(NgAggArgs a) {
    constexpr int LPW = 64 / G;                               // lines per wave
    constexpr int LS = 68;                                    // LDS stride of one array: 64 entries + 4 (a multiple of 4: 16-byte reads)
    __shared__ __attribute__((aligned(16))) uint32_t sC[4 * LPW][2][3][LS];    // [line of the workgroup][buffer][key, cost & 0xFF, (cost + P1) & 0xFF][place]
    __shared__ uint32_t sPick;
    if (ng_agg_not_mine(a, &sPick)) return;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.kstat[257] = 1u;   // tells the WTA where the sums are (L4, not S)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / G, pl_ = lane % G;                 // line of the wave, first place in its list
    int bb = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if ((int)blockIdx.x >= a.blk_begin_c[i]) bb = i;
    const int slot = a.slot_of_c[bb];
    const int base = slot & 1;                               // 0: along x, 1: along y
    const bool mirror = slot >= 2;
    const int W = a.W, H = a.H, D = a.D;
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    const int line0 = (((int)blockIdx.x - a.blk_begin_c[bb]) * 4 + wave) * LPW;
    if (line0 >= nlines) return;                              // wave-uniform; the waves of a workgroup never meet again
#if FSGM_NG_PRIO
    // the lines along the longer side are the launch's critical path (1242 serial steps against 375): their waves issue first
    // whenever a SIMD has the choice, the short lines' waves fill what is left
    if (len > nlines) __builtin_amdgcn_s_setprio(3);
#endif
    const bool lact = line0 + sub < nlines;                   // lines past the last redo the last one without adding to S
    const int line = min(line0 + sub, nlines - 1);
    const size_t f = blockIdx.y;
    const uint32_t* __restrict__ ckf = a.ck + f * (size_t)NP * D;
    const uint16_t* __restrict__ cmf = a.cm + f * (size_t)NP * D;
    const uint8_t* __restrict__ dkf = a.dk + f * (size_t)NP;
    int16_t* __restrict__ L4f = a.L4 + f * (size_t)NP * NG_L4_PER_PIXEL + slot * 64;
    int pix0 = base == 0 ? line * W : line;
    if (mirror) pix0 = NP - 1 - pix0;
    const int dpix = (mirror ? -1 : 1) * (base == 0 ? 1 : W);
    const uint32_t dent = (uint32_t)(dpix * D);
    const uint32_t pl = (uint32_t)min(pl_, D - 1);           // lanes past the list read inside the pixel's D slots
    uint32_t p_cur = (uint32_t)pix0, p_fet = p_cur, e_fet = p_cur * (uint32_t)D;
    constexpr int PF = FSGM_NG_CPF;
    // entry loads through 32-bit byte offsets from the frame's (uniform) base: one add-shift each instead of a 64-bit address
    auto ld_key = [&](const uint32_t e) -> uint32_t { return *(const uint32_t*)((const char*)ckf + (size_t)(uint32_t)(e << 2)); };
    auto ld_meta = [&](const uint32_t e) -> uint32_t { return *(const uint16_t*)((const char*)cmf + (size_t)(uint32_t)(e << 1)); };
    uint32_t rkey[PF], rmeta[PF], rlen[PF];                   // the first round's entry of the coming steps
    // one line a wave: the same entries once more as the four rows see them when the lists are short (lane & 15: every row the
    // candidates 0 .. 15; ng_match4_key_rows)
    uint32_t rkey16[G == 64 ? PF : 1], rmeta16[G == 64 ? PF : 1];
    const uint32_t pl16 = (uint32_t)min(lane & 15, D - 1);
    const int row = lane >> 4;
#pragma unroll
    for (int k = 0; k < PF; k++) {
        rkey[k] = ld_key(e_fet + pl); rmeta[k] = ld_meta(e_fet + pl); rlen[k] = dkf[p_fet];
        if constexpr (G == 64) { rkey16[k] = ld_key(e_fet + pl16); rmeta16[k] = ld_meta(e_fet + pl16); }
        if (k + 1 < len) { p_fet += (uint32_t)dpix; e_fet += dent; }
    }
    uint32_t* b0 = &sC[wave * LPW + sub][0][0][0];
    uint32_t* b1 = &sC[wave * LPW + sub][1][0][0];
    uint32_t m = 0;                                           // :172 / :77: stored minimum 0 at a path start
    int K4pre = 0;                                            // (wave-uniform: the longest staged list of the wave's lines)
    // one step; its ring slot is named by the caller, so that the steady-state loop below is straight-line code with the
    // slots at fixed registers: with a way out of the middle of the unrolled group the compiler rotates the ring through
    // copies, and every copy waits for the loads just requested (and for the step's atomic add) -- vmcnt(0) per step
    auto step = [&](const int t, uint32_t& rk, uint32_t& rm, uint32_t& rl, uint32_t& rk16, uint32_t& rm16) {
        uint32_t key = rk, meta = rm;
        // one line a wave: the list length is the same in every lane -- as a scalar, so that everything decided by it is a scalar
        // branch (as a vector value the compiler masks exec around both sides of each decision)
        const int K = G == 64 ? (int)__builtin_amdgcn_readfirstlane(rl) : (int)rl;
        bool rows16 = false;                                  // wave-uniform: this step's and the previous step's lists fit 16 lanes
        if constexpr (G == 64) {
            rows16 = K <= 16 && K4pre <= 16 && t > 0;
            if (rows16) { key = rk16; meta = rm16; }
            rk16 = ld_key(e_fet + pl16); rm16 = ld_meta(e_fet + pl16);
        }
        rk = ld_key(e_fet + pl); rm = ld_meta(e_fet + pl); rl = dkf[p_fet];
        if (t + PF + 1 < len) { p_fet += (uint32_t)dpix; e_fet += dent; }
        int kmax = K;                                         // the longest list of the wave's lines at this step (K is uniform inside a line)
        if (LPW > 1) {                                        // K is uniform inside a line's lanes: the rows' values through row swaps
            // (two __shfl_xor = two dependent trips through the LDS crossbar per step before round 4)
            uint32_t a_ = (uint32_t)kmax, b_ = a_;
            if (G <= 16) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a_), "+v"(b_)); a_ = max(a_, b_); b_ = a_; }
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a_), "+v"(b_));
            kmax = (int)max(a_, b_);
        }
        const uint32_t jump = (m + (uint32_t)a.P2) & 0xFF;
        const uint32_t ecur = p_cur * (uint32_t)D;
        uint32_t lov = 0xFFFFFFFFu;
        // one round = G places of the list.  The first uses the prefetched entry; further rounds (a list longer than G: rare, and
        // wave-uniform) load theirs on the spot, in a loop of their own -- with the loads inside the first round's code path the
        // compiler waited for ALL outstanding memory operations (vmcnt(0): the prefetches just issued and the last step's store)
        // on every step of every line
        auto round = [&](const int e0, const uint32_t key_, const uint32_t meta_) {
            const int e = e0 + pl_;
            const bool act = e < K;
            const int cost = (int)(meta_ & 0xFFu);
            int o = cost;
            if (G == 64 && rows16) o = (cost + (int)ng_match4_key_rows(b0, b0 + LS, b0 + 2 * LS, K4pre, key_, jump, row)) - (int)m;
            else if (t > 0) o = (cost + (int)ng_match4_key(b0, b0 + LS, b0 + 2 * LS, K4pre, key_, jump)) - (int)m;
            if constexpr (G == 64) {
                // every lane writes its slot (e < 64 < LS), entries or neutral ones: selects instead of two masked paths
                b1[e] = act ? key_ : NG_PADKEY;
                b1[LS + e] = act ? ((uint32_t)o & 0xFF) : 0xFFFFu;
                b1[2 * LS + e] = act ? ((uint32_t)(o + a.P1) & 0xFF) : 0xFFFFu;
                if (act) L4f[p_cur * (uint32_t)NG_L4_PER_PIXEL + (uint32_t)e] = (int16_t)o;    // :249 (|o| <= 510; summed by the WTA); one line a wave: lact
                // :74 narrowed.  With the rows working for the same 16 candidates every row takes part: each row's four DPP stages
                // then end with the line's minimum in every lane, no trip through scalar registers
                const bool actm = rows16 ? (lane & 15) < K : act;
                lov = actm ? min(lov, (uint32_t)o & 0xFFu) : lov;
            } else if (act) {
                b1[e] = key_; b1[LS + e] = (uint32_t)o & 0xFF; b1[2 * LS + e] = (uint32_t)(o + a.P1) & 0xFF;
                if (lact) L4f[p_cur * (uint32_t)NG_L4_PER_PIXEL + (uint32_t)e] = (int16_t)o;   // :249 (|o| <= 510; summed by the WTA)
                lov = min(lov, (uint32_t)o & 0xFFu);                                           // :74 narrowed
            } else if (e < LS) {                              // neutral entries in every other slot of the round
                b1[e] = NG_PADKEY; b1[LS + e] = 0xFFFFu; b1[2 * LS + e] = 0xFFFFu;
            }
        };
        round(0, key, meta);
        if constexpr (G < 64) {
            for (int e0 = G; e0 < kmax; e0 += G) {            // wave-uniform bound
                // These two loads are spelled in asm, with their own wait, so that the compiler's wait-count bookkeeping never
                // sees a memory operation inside the step: with a load on a conditional path it falls back to waiting for
                // everything outstanding at the top of EVERY step (vmcnt(3) right behind the three prefetches: the last step's
                // store and 21 loads that need not be back for another seven steps).
                const uint32_t ee = (uint32_t)min(e0 + pl_, D - 1);
                const uint32_t* pk = ckf + (ecur + ee);
                const uint16_t* pm = cmf + (ecur + ee);
                uint32_t k2, m2;
                asm volatile("global_load_dword %0, %2, off\n\tglobal_load_ushort %1, %3, off\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(k2), "=&v"(m2) : "v"(pk), "v"(pm) : "memory");
                round(e0, k2, m2);
            }
        }
        uint32_t lo;
        if constexpr (G == 64) lo = rows16 ? group_min_u32<16>(lov) : wave_min_u32(lov); else lo = group_min_u32<G>(lov);
        m = t > 0 ? lo : 0u;
        K4pre = min((kmax + 3) & ~3, 64);
        p_cur += (uint32_t)dpix;
        __builtin_amdgcn_wave_barrier();
        uint32_t* tmp = b0; b0 = b1; b1 = tmp;
    };
    int t0 = 0;
    for (; t0 + PF <= len; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) step(t0 + u, rkey[u], rmeta[u], rlen[u], rkey16[G == 64 ? u : 0], rmeta16[G == 64 ? u : 0]);
    }
#pragma unroll
    for (int u = 0; u < PF - 1; u++)
        if (t0 + u < len) step(t0 + u, rkey[u], rmeta[u], rlen[u], rkey16[G == 64 ? u : 0], rmeta16[G == 64 ? u : 0]);   // wave-uniform
}

// S of every member of a group of repeats := S of the group's first member (for reading S back: the compact kernel
// adds there only)
__global__ __launch_bounds__(256) void ng_fill_repeats_kernel(uint32_t* __restrict__ S, const uint16_t* __restrict__ dd, const uint16_t* __restrict__ cm,
                                                              long long n, int D) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long p0 = i - i % D;
    const long long first = p0 + (cm[p0 + (dd[i] & 0x7FFFu)] >> 8);     // the group's first member holds the sums
    if (first != i) S[i] = S[first];
}

// The matcher of one candidate over the entries [e0, e1) of a staged predecessor (e0 % 4 == 0): the last exact
// match (NG_NOMATCH if none) and the minimum over the near entries (NG_BIG if none), to be folded in entry order.
constexpr uint32_t NG_NOMATCH = 0xFFFFFFFFu, NG_BIG = 0xFFFFu;
__device__ __forceinline__ void ng_match_range(const NgPre& q, int e0, int e1, int mvx, int mvy, bool safe, uint32_t& m1, uint32_t& m2) {
    if (safe) {
        const uint32_t ck2 = ng_pack_mv(mvx, mvy) + 0x00020002u;
        const uint32_t* qk = (const uint32_t*)q.x;
        int d2 = e0;
        for (; d2 + 4 <= e1; d2 += 4) {
            const uint4 k4 = *(const uint4*)(qk + d2);
            const uint4 c8 = *(const uint4*)(q.c8 + d2), cp = *(const uint4*)(q.cp + d2);
            const uint32_t ka[4] = {k4.x, k4.y, k4.z, k4.w};
            const uint32_t c8a[4] = {c8.x, c8.y, c8.z, c8.w}, cpa[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t t = pk_sub(ck2, ka[i]);
                const bool nr = pk_min(t, 0x00040004u) == t, eq = t == 0x00020002u;
                m1 = eq ? c8a[i] : m1;                                            // last match wins
                uint32_t sel = nr ? cpa[i] : NG_BIG;
                sel = eq ? NG_BIG : sel;
                m2 = min(m2, sel);
            }
        }
        for (; d2 < e1; d2++) {
            const uint32_t t = pk_sub(ck2, qk[d2]);
            const bool nr = pk_min(t, 0x00040004u) == t, eq = t == 0x00020002u;
            m1 = eq ? q.c8[d2] : m1;
            if (nr && !eq) m2 = min(m2, q.cp[d2]);
        }
    } else {
        for (int d2 = e0; d2 < e1; d2++) {
            const int qx = q.x[d2], qy = q.y[d2];
            if (mvx == qx && mvy == qy) m1 = q.c8[d2];
            else if (near2(mvx, qx) && near2(mvy, qy)) m2 = min(m2, q.cp[d2]);
        }
    }
}

// ng_agg_lines_kernel with the matcher of every candidate cut PARTS ways over the predecessor's entries:
// 256 * PARTS threads advance the same 256/D lines, a step is shorter by that factor and costs a second barrier.  For a single
// frame the long lines (1242 steps against 375) outlive the short ones at about one wave per SIMD, where the
// step latency is all that counts; with the long lines' blocks launched first (slot_of) the short ones fill in.
template <int PARTS>
__global__ __launch_bounds__(256 * PARTS) void ng_agg_split_kernel(NgAggArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sNg[];
    if (ng_agg_not_mine(a, sNg)) return;
    // [line][2 buffers][4 arrays][Dp] | [line][8]: 3 minima, staged lengths at [4 + step parity] | [line][Dp] candidate mvx | [line][Dp] mvy | [PARTS - 1][line][Dp][2]
    int k = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if ((int)blockIdx.x >= a.blk_begin[i]) k = i;
    const int slot = a.slot_of[k];
    const int base = slot & 1;                               // 0: along x, 1: along y
    const bool mirror = slot >= 2;
    const int W = a.W, H = a.H, D = a.D, Dp = (D + 3) & ~3;
    const int LPB = 256 / D;                                  // lines per workgroup
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    const int part = threadIdx.x >> 8, g = threadIdx.x & 255;
    const int ll = min(g / D, LPB - 1), cand = g - (g / D) * D;
    const bool tact = g < LPB * D;                            // this thread holds a candidate
    const int line = ((int)blockIdx.x - a.blk_begin[k]) * LPB + ll;
    const bool lact = tact && line < nlines;
    const int linec = min(line, nlines - 1);
    const size_t f = blockIdx.y;
    const Cand* __restrict__ Cf = a.C + f * (size_t)NP * D;
    uint32_t* __restrict__ Sf = a.S + f * (size_t)NP * D;
    uint32_t* buf0 = sNg + (size_t)ll * 8 * Dp;
    uint32_t* buf1 = buf0 + 4 * Dp;
    uint32_t* smin = sNg + (size_t)LPB * 8 * Dp + ll * 8;
    uint32_t* cxs = sNg + (size_t)LPB * (8 * Dp + 8) + (size_t)ll * Dp;
    uint32_t* cys = cxs + (size_t)LPB * Dp;
    uint32_t* parts = sNg + (size_t)LPB * (10 * Dp + 8);      // [(part - 1) * LPB + ll][Dp][2]
    if (part == 0 && cand < 3 && tact) smin[cand] = 255u;
    const bool safe = *a.unsafe == 0;
    const uint16_t* __restrict__ ddf = a.dd ? a.dd + f * (size_t)NP * D : nullptr;
    const uint8_t* __restrict__ dkf = a.dk ? a.dk + f * (size_t)NP : nullptr;
    auto pix_of = [&](int t) {
        int x = base == 0 ? t : linec, y = base == 0 ? linec : t;
        if (mirror) { x = W - 1 - x; y = H - 1 - y; }
        return (size_t)y * W + x;
    };
    constexpr int PF = 4;
    Cand ring[PF];                                            // part 0 only: the candidates of steps t .. t+3,
    uint32_t rpl[PF], rlen[PF];                               // their places in the pixel's list without repeats, its length
    if (part == 0) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const size_t px = pix_of(min(i, len - 1));
            ring[i] = Cf[px * D + cand];
            rpl[i] = ddf ? ddf[px * D + cand] : (uint32_t)cand;
            rlen[i] = dkf ? dkf[px] : (uint32_t)D;
        }
    }
    __syncthreads();
    for (int t0 = 0; t0 < len; t0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int t = t0 + u;
        if (t >= len) break;                                  // block-uniform
        const NgPre q{(const int32_t*)buf0, (const int32_t*)buf0 + Dp, buf0 + 2 * Dp, buf0 + 3 * Dp};
        // ---- phase 1: partial matchers
        uint32_t m1 = NG_NOMATCH, m2 = NG_BIG;
        if (t > 0) {
            const int Kpre = (int)smin[4 + ((t - 1) & 1)], Kq = safe ? (Kpre + 3) & ~3 : Kpre;   // safe: neutral entries fill the last block
            const int pb = (((Kpre + 3) >> 2) + PARTS - 1) / PARTS;
            const int e0 = min(Kq, 4 * pb * part), e1 = min(Kq, 4 * pb * (part + 1));
            ng_match_range(q, e0, e1, (int)cxs[cand], (int)cys[cand], safe, m1, m2);
            if (part > 0) {
                uint32_t* pr = parts + (((size_t)(part - 1) * LPB + ll) * Dp + cand) * 2;
                pr[0] = m1; pr[1] = m2;
            }
        }
        __syncthreads();
        // ---- phase 2: fold in entry order, finish the step, stage it for the next one
        if (part == 0) {
            const Cand c = ring[u];
            const uint32_t place = rpl[u], K = rlen[u];
            {
                const size_t px = pix_of(min(t + PF, len - 1));
                ring[u] = Cf[px * D + cand];
                rpl[u] = ddf ? ddf[px * D + cand] : (uint32_t)cand;
                rlen[u] = dkf ? dkf[px] : (uint32_t)D;
            }
            const size_t off = pix_of(t) * D;
            const uint32_t m = t >= 2 ? smin[(t - 1) % 3] : 0u;    // :172 / :77; stored minimum 0 at a path start
            const uint32_t jump = (m + (uint32_t)a.P2) & 0xFF;
            int o = c.cost;
            if (t > 0) {
#pragma unroll
                for (int e = 0; e < PARTS - 1; e++) {
                    const uint32_t* pr = parts + (((size_t)e * LPB + ll) * Dp + cand) * 2;
                    m1 = pr[0] != NG_NOMATCH ? pr[0] : m1;
                    m2 = min(m2, pr[1]);
                }
                const uint32_t best = min(jump, min(m1 == NG_NOMATCH ? jump : m1, m2));
                o = (c.cost + (int)best) - (int)m;
                if (tact) atomicMin(&smin[t % 3], (uint32_t)o & 0xFF);            // :74 narrowed
            }
            if (tact) {
                if (place < 0x8000u) {                                    // a kept entry (repeats carry 0x8000 | the place of the entry they repeat)
                    buf1[place] = safe ? ng_pack_mv(c.mvx, c.mvy) : (uint32_t)c.mvx; buf1[Dp + place] = (uint32_t)c.mvy;
                    buf1[2 * Dp + place] = (uint32_t)o & 0xFF; buf1[3 * Dp + place] = (uint32_t)(o + a.P1) & 0xFF;
                }
                if (cand < 3 && K + cand < (uint32_t)Dp) {     // neutral entries up to the next multiple of 4
                    buf1[K + cand] = NG_PADKEY; buf1[Dp + K + cand] = 0x7FFFFFFFu;
                    buf1[2 * Dp + K + cand] = 0xFFFFu; buf1[3 * Dp + K + cand] = 0xFFFFu;
                }
                if (lact) atomicAdd(&Sf[off + cand], (uint32_t)o);                // :249
                if (cand == 0) { smin[(t + 1) % 3] = 255u; smin[4 + (t & 1)] = K; }
                const Cand cn = ring[(u + 1) % PF];                               // the next step's candidate, for all parts
                cxs[cand] = (uint32_t)cn.mvx; cys[cand] = (uint32_t)cn.mvy;
            }
        }
        __syncthreads();
        uint32_t* tmp = buf0; buf0 = buf1; buf1 = tmp;
      }
    }
}

// WTA -> winning candidate's motion vector (calc_pyd_cost_sgm_ng.cpp:281-299); one wave per pixel.  With the kept-entry
// table of launch_ng_dedupe the search runs over the K groups of repeats instead of the D candidates: a group's sum sits
// at its first member's index, every member ties with it, so the first minimum over d is the smallest (sum, first index).
__global__ __launch_bounds__(256) void ng_wta_kernel(NgWtaArgs a) {
    // 16 lanes a pixel, 4 pixels a wave: the kept lists are mostly shorter than 16, and D = 81 candidates take 6 rounds
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, l16 = lane & 15;
    const int NP = a.W * a.H, D = a.D;
    const int pw = (blockIdx.x * 4 + wave) * 4;
    if (pw >= NP) return;                                     // wave-uniform
    const bool pact = pw + sub < NP;
    const int p = min(pw + sub, NP - 1);
    const size_t f = blockIdx.y;
    const uint32_t* Sp = a.S + f * (size_t)NP * D + (size_t)p * D;
    const uint16_t* cmp = a.cm ? a.cm + f * (size_t)NP * D + (size_t)p * D : nullptr;
    const int n = cmp ? (int)a.dk[f * (size_t)NP + p] : D;
    const bool l4 = cmp && a.L4 && a.kstat && a.kstat[257] != 0;              // launch-uniform: the compact kernel ran
    const int16_t* Lp = a.L4 + (f * (size_t)NP + p) * NG_L4_PER_PIXEL;
    uint32_t lo = 0xFFFFFFFFu, idx = 0xFFFFFFFFu;
    for (int e = l16; e < n; e += 16) {
        const uint32_t d = cmp ? (uint32_t)(cmp[e] >> 8) : (uint32_t)e;
        const uint32_t s = l4 ? (uint32_t)((int)Lp[e] + (int)Lp[64 + e] + (int)Lp[128 + e] + (int)Lp[192 + e]) : Sp[d];
        if (s < lo || idx == 0xFFFFFFFFu || (s == lo && d < idx)) { lo = s; idx = d; }
    }
    uint32_t glo = lo;
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) glo = min(glo, (uint32_t)__shfl_xor((int)glo, s));
    uint32_t gidx = (lo == glo && idx != 0xFFFFFFFFu) ? idx : 0xFFFFFFFFu;
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) gidx = min(gidx, (uint32_t)__shfl_xor((int)gidx, s));
    if (l16 == 0 && pact) {
        // the winner's motion vector: from its 4-byte key where the launch kept those (compact matcher, every key in range:
        // the 12-byte list does not exist then), from the Cand list otherwise
        const bool k4 = l4 && a.K4 && a.flags && a.flags[1] == 0;
        const Cand c = k4 ? ng_unkey4(a.K4[f * (size_t)NP * D + (size_t)p * D + gidx]) : a.C[f * (size_t)NP * D + (size_t)p * D + gidx];
        a.minC[f * NP + p] = glo;
        a.flow[f * 2 * (size_t)NP + p] = (double)c.mvx;
        a.flow[f * 2 * (size_t)NP + NP + p] = (double)c.mvy;
    }
}

// sub-pixel on raw single-pixel census costs (calc_pyd_cost_sgm_ng.cpp:308-368)
__device__ __forceinline__ double ng_parab(double cl, double c0, double cr) {
    return cr < cl ? __ddiv_rn(__ddiv_rn(__dsub_rn(cr, cl), __dsub_rn(c0, cl)), 2.0)
                   : __ddiv_rn(__ddiv_rn(__dsub_rn(cr, cl), __dsub_rn(c0, cr)), 2.0);
}

__global__ __launch_bounds__(256) void ng_subpixel_kernel(NgSubpixArgs a) {
    const int W = a.W, H = a.H, NP = W * H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const size_t f = blockIdx.y;
    const int y = p / W, x = p - y * W;
    double* fx = a.flow + f * 2 * (size_t)NP;
    double* fy = fx + NP;
    const uint32_t* cen2 = a.cen2 + f * (size_t)NP;
    const uint32_t c1 = a.cen1[f * (size_t)NP + p];
    const int tx = f64_to_i32_x86(__dadd_rn(fx[p], (double)x)), ty = f64_to_i32_x86(__dadd_rn(fy[p], (double)y));   // :325-326
    if (!(tx > 1 && tx < W - 1 && ty > 1 && ty < H - 1)) return;                                                  // :328
    const double c0 = (double)__popc(c1 ^ cen2[(size_t)ty * W + tx]);
    double cl = (double)__popc(c1 ^ cen2[(size_t)ty * W + tx - 1]);
    double cr = (double)__popc(c1 ^ cen2[(size_t)ty * W + tx + 1]);
    if (c0 >= cl || c0 >= cr) return;                                                                             // :337
    fx[p] = __dadd_rn(fx[p], ng_parab(cl, c0, cr));
    cl = (double)__popc(c1 ^ cen2[(size_t)(ty - 1) * W + tx]);
    cr = (double)__popc(c1 ^ cen2[(size_t)(ty + 1) * W + tx]);
    if (c0 >= cl || c0 >= cr) return;                                                                             // :354
    fy[p] = __dadd_rn(fy[p], ng_parab(cl, c0, cr));
}

// =============================================================================================
// on-the-fly variant (calc_cost_sgm_ng.cpp:188-419).  Raster-serial by construction: the
// candidates of pixel (x,y) are the two best motion vectors left in the path buffers that are
// about to be overwritten -- L1's from two pixels earlier in raster order, L2/L3/L4's from two
// rows earlier (:276-277) -- plus one libc-rand() hint per buffer (:148-149), each expanded 3x3.
// One workgroup walks one frame pixel by pixel; frames of a batch run on different CUs.
// otf_kernel (W < 4, and the plain statement of the step): 256 threads, the 108 candidate costs spread
// over the threads, then wave k runs path k's O(D^2) matcher + top-2 tracking, four barriers per pixel.
// otf_pipe_kernel (below) is the form that runs on real images.
// =============================================================================================
// the two best entries and the top-N slots, by the wave that holds all 108 new costs.
// Top-2 by insertion in ascending d with strict '<' (:84-96): entry j precedes entry k when cost_j < cost_k,
// or cost_j == cost_k and j < k.  Slots start at cost 255 (:54-55) and an entry that is not < 255 never
// enters; the motion vectors of untouched slots stay as they are; the first insertion shifts old slot 0
// (cost 255, old mv) down into slot 1 (:92-95).  key = (cost + 2^16) << 8 | d gives exactly that order: a cost
// is (candidate cost <= 25) + (best <= 255) - (m <= 255).
__device__ __forceinline__ void otf_keep_best(Cand* Lout, int costa, int costb, bool has1, int lane) {
    const uint32_t KMAX = 0xFFFFFFFFu;
    const uint32_t k0 = ((uint32_t)(costa + 65536) << 8) | (uint32_t)lane;
    const uint32_t k1 = has1 ? ((uint32_t)(costb + 65536) << 8) | (uint32_t)(lane + 64) : KMAX;
    const uint32_t best = wave_min_u32(min(k0, k1));
    const uint32_t second = wave_min_u32(min(k0 == best ? KMAX : k0, k1 == best ? KMAX : k1));
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        Cand t0 = Lout[OTF_D], t1 = Lout[OTF_D + 1];
        t0.cost = 255; t1.cost = 255;
        const int bc = (int)(best >> 8) - 65536, bd = (int)(best & 0xFF);
        const int sc = (int)(second >> 8) - 65536, sd = (int)(second & 0xFF);
        if (bc < 255) {
            t1 = t0;
            t0 = Lout[bd];
            if (second != KMAX && sc < 255) t1 = Lout[sd];
        }
        Lout[OTF_D] = t0;
        Lout[OTF_D + 1] = t1;
    }
}

__device__ __forceinline__ void otf_step_wave(Cand* Lout, const NgPre& pre, uint32_t m, const Cand* Cc, int lane,
                                              int P2, bool safe) {
    // calc_cost_sgm_ng.cpp:46-98 for one path, executed by one wave (64 lanes over 108 candidates, two per lane);
    // pre: the predecessor's entries as four arrays, m: its stored minimum (:53)
    const uint32_t jump = (m + (uint32_t)P2) & 0xFF;
    const bool has1 = lane + 64 < OTF_D;
    const Cand ca = Cc[lane], cb = Cc[has1 ? lane + 64 : lane];
    int costa, costb;
    if (safe) {
        uint32_t ba, bb;
        ng_match4_pair(pre, OTF_D, ca.mvx, ca.mvy, cb.mvx, cb.mvy, jump, ba, bb);
        costa = (ca.cost + (int)ba) - (int)m;
        costb = (cb.cost + (int)bb) - (int)m;
    } else {
        costa = ng_match4(pre, OTF_D, ca.mvx, ca.mvy, ca.cost, m, jump, false);
        costb = ng_match4(pre, OTF_D, cb.mvx, cb.mvy, cb.cost, m, jump, false);
    }
    Cand oa = ca, ob = cb;
    oa.cost = costa; ob.cost = costb;
    Lout[lane] = oa;
    if (has1) Lout[lane + 64] = ob;
    otf_keep_best(Lout, costa, costb, has1, lane);
}

__global__ __launch_bounds__(256) void otf_kernel(OtfArgs a) {
    __shared__ Cand sC[OTF_D];                    // candidates of the current pixel
    __shared__ Cand sL1[2][OTF_E];                // L1 double buffer (:197)
    __shared__ __attribute__((aligned(16))) uint32_t sPre[4][4][OTF_D];        // predecessor entries staged per path: mvx, mvy, cost & 0xFF, (cost + P1) & 0xFF
    __shared__ int sUnsafe;                       // sticky: a motion vector near the int range was seen (exact matcher from then on)
    __shared__ Cand sOut[4][OTF_E];               // new entries per path
    __shared__ int sHint[4][3][2];                // [buffer][hint][mvx,mvy]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = a.W, H = a.H, NP = W * H;
    const size_t f = blockIdx.x;
    const uint8_t* If = a.I1 + f * (size_t)NP;
    const uint32_t* cen1 = a.cen1 + f * (size_t)NP;
    const uint32_t* cen2 = a.cen2 + f * (size_t)NP;
    const int32_t* rnd = a.rnd + f * (size_t)NP * 8;
    const size_t rowE = (size_t)W * OTF_E;
    Cand* Lrow = a.Lrow + f * 6 * rowE;           // [L2,L3,L4][2][W][OTF_E]
    for (int i = tid; i < 2 * OTF_E; i += 256) { Cand z = {0, 0, 0}; sL1[i / OTF_E][i % OTF_E] = z; }   // :204
    if (tid == 0) sUnsafe = a.exact;
    __syncthreads();
    int l1cur = 1, rowcur = 1;                    // :248, :250-254
    for (int y = 0; y < H; y++) {
        const int rowpre = rowcur ^ 1;
        for (int x = 0; x < W; x++) {
            const int p = y * W + x;
            const int l1pre = l1cur ^ 1;
            Cand* L2c = Lrow + (0 * 2 + rowcur) * rowE + (size_t)x * OTF_E;
            Cand* L3c = Lrow + (1 * 2 + rowcur) * rowE + (size_t)x * OTF_E;
            Cand* L4c = Lrow + (2 * 2 + rowcur) * rowE + (size_t)x * OTF_E;
            // ---- hints: top-N entries of the buffers about to be overwritten (:276-277) + rand
            if (tid < 12) {
                const int l = tid / 3, i = tid % 3;
                int mvx, mvy;
                if (i < 2) {
                    const Cand* src = l == 0 ? &sL1[l1cur][OTF_D] : (l == 1 ? L2c + OTF_D : (l == 2 ? L3c + OTF_D : L4c + OTF_D));
                    mvx = src[i].mvx; mvy = src[i].mvy;
                } else {
                    mvx = rnd[(size_t)p * 8 + 2 * l] % 256 - 128;                // :148
                    mvy = rnd[(size_t)p * 8 + 2 * l + 1] % 128 - 64;             // :149
                }
                sHint[l][i][0] = mvx; sHint[l][i][1] = mvy;
            }
            __syncthreads();
            // ---- candidate costs (:122-186): clamp-border 5x5 mean of Hamming costs
            if (tid < OTF_D) {
                const int hi = tid / 9, k = tid % 9;
                const int offy = k / 3 - 1, offx = k % 3 - 1;                    // :153-154 offy outer
                const int mvx = sHint[hi / 3][hi % 3][0], mvy = sHint[hi / 3][hi % 3][1];
                uint32_t sum = 0;
#pragma unroll
                for (int ay = -2; ay <= 2; ay++) {
                    const int y1 = clampi(y + ay, 0, H - 1);
                    const int y2 = clampi((offy + y1) + mvy, 0, H - 1);
#pragma unroll
                    for (int ax = -2; ax <= 2; ax++) {
                        const int x1 = clampi(x + ax, 0, W - 1);
                        const int x2 = clampi((offx + x1) + mvx, 0, W - 1);
                        sum += __popc(cen1[(size_t)W * y1 + x1] ^ cen2[(size_t)W * y2 + x2]);
                    }
                }
                Cand c;
                c.cost = f64_to_i32_x86(__dadd_rn(__ddiv_rn(__dmul_rn(1.0, (double)sum), 25.0), 0.5));   // :177
                c.mvx = mvx + offx; c.mvy = mvy + offy;
                sC[tid] = c;
                if (!(c.mvx > -(1 << 30) && c.mvx < (1 << 30) && c.mvy > -(1 << 30) && c.mvy < (1 << 30))) sUnsafe = 1;
            }
            __syncthreads();
            // ---- per path (wave k = path buffer k): start copy or matcher step
            {
                const bool s1 = x == 0, s2 = x == 0 || y == 0, s3 = y == 0, s4 = y == 0 || x == W - 1;
                const bool is_start = wave == 0 ? s1 : (wave == 1 ? s2 : (wave == 2 ? s3 : s4));
                Cand* out = sOut[wave];
                // current content of the slot being overwritten (its top-N mvs survive a start copy)
                const Cand* curbuf = wave == 0 ? sL1[l1cur] : (wave == 1 ? L2c : (wave == 2 ? L3c : L4c));
                if (lane < 2) out[OTF_D + lane] = curbuf[OTF_D + lane];
                __builtin_amdgcn_wave_barrier();
                if (is_start) {
                    for (int d = lane; d < OTF_D; d += 64) out[d] = sC[d];      // :280,284,290,294,297,304
                    if (lane == 0) out[OTF_D].cost = 0;
                } else {
                    const Cand* psrc;
                    int pp;
                    if (wave == 0) { psrc = sL1[l1pre]; pp = If[p - 1]; }
                    else if (wave == 1) { psrc = Lrow + (0 * 2 + rowpre) * rowE + (size_t)(x - 1) * OTF_E; pp = If[p - W - 1]; }
                    else if (wave == 2) { psrc = Lrow + (1 * 2 + rowpre) * rowE + (size_t)x * OTF_E; pp = If[p - W]; }
                    else { psrc = Lrow + (2 * 2 + rowpre) * rowE + (size_t)(x + 1) * OTF_E; pp = If[p - W + 1]; }
                    uint32_t (*pre)[OTF_D] = sPre[wave];
                    for (int d = lane; d < OTF_D; d += 64) {
                        const Cand e = psrc[d];
                        pre[0][d] = (uint32_t)e.mvx; pre[1][d] = (uint32_t)e.mvy;
                        pre[2][d] = (uint32_t)e.cost & 0xFF; pre[3][d] = (uint32_t)(e.cost + a.P1) & 0xFF;
                    }
                    const uint32_t m = (uint32_t)psrc[OTF_D].cost & 0xFF;        // :53
                    __builtin_amdgcn_wave_barrier();
                    const int P2 = abs((int)If[p] - pp) > 50 ? a.P2 / 8 : a.P2;  // :101-105 adaptive P2
                    const NgPre q{(const int32_t*)pre[0], (const int32_t*)pre[1], pre[2], pre[3]};
                    otf_step_wave(out, q, m, sC, lane, P2, sUnsafe == 0);
                }
            }
            __syncthreads();
            // ---- S = L1+L3+L2+L4 (:357-362) and WTA for this pixel (:389-407); wave 0 only
            if (wave == 0) {
                unsigned long long key = ~0ull;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int d = lane + 64 * i;
                    if (d < OTF_D) {
                        const uint32_t s = (uint32_t)(sOut[0][d].cost + sOut[2][d].cost) + (uint32_t)(sOut[1][d].cost + sOut[3][d].cost);
                        const unsigned long long k = ((unsigned long long)s << 8) | (unsigned)d;
                        key = k < key ? k : key;
                    }
                }
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) {
                    const unsigned long long o = __shfl_xor(key, s);
                    key = o < key ? o : key;
                }
                if (lane == 0) {
                    const int idx = (int)(key & 0xFF);
                    a.minC[f * NP + p] = (uint32_t)(key >> 8);
                    a.flow[f * 2 * (size_t)NP + p] = (double)sC[idx].mvx;
                    a.flow[f * 2 * (size_t)NP + NP + p] = (double)sC[idx].mvy;
                }
            }
            // ---- write the new entries back into the buffers
            {
                Cand* dst = wave == 0 ? sL1[l1cur] : (wave == 1 ? L2c : (wave == 2 ? L3c : L4c));
                for (int d = lane; d < OTF_E; d += 64) dst[d] = sOut[wave][d];
            }
            __threadfence_block();
            __syncthreads();
            l1cur ^= 1;                                                          // :365-367
        }
        rowcur ^= 1;                                                             // :371-384
    }
}

// =============================================================================================
// The same recursion, software-pipelined over raster order (W >= 4), 14 waves per frame.
// Everything pixel n+1 needs from global memory was written at least two pixels earlier -- the hints come
// from buffers last written at pixel n-1 (L1, in LDS) or two rows back, the predecessors of the three
// downward paths from the row above -- so no global round trip is left on the critical path:
//   waves 0..11: path p = wave / 3, part e = wave % 3.  The O(D^2) matcher of pixel n is cut three ways
//     over the predecessor's entries (36 each, fetched into registers during pixel n-1, staged into LDS by
//     the wave that scans them); every lane carries two candidates through its part and leaves
//     (last exact match, minimum over the near entries); after one barrier part 0 folds the three partial
//     results in entry order, finishes the costs, keeps the two best and writes the path's buffer back.
//   waves 12..13: the 108 candidate costs of pixel n+1 (hints fetched during pixel n-1 as well), then the
//     minimum over the four paths for pixel n-1.
// Motion vectors within +-0x3FF0 (anything a real image produces) are matched as packed 2 x u16 keys:
// t = (c + (2,2)) - q per half; equal means t == (2,2), near means both halves <= 4 -- 8 VALU per
// entry and candidate, no scalar mask arithmetic.  A vector outside that range switches the frame to
// the exact form for good (sUnsafe).
// =============================================================================================
constexpr uint32_t OTF_NOMATCH = 0xFFFFFFFFu, OTF_BIG = 0xFFFFu;
constexpr int OTF_PART = 36;                      // entries per part: 9 blocks of 4

__device__ __forceinline__ void otf_match_part(const uint32_t* __restrict__ key, const uint32_t* __restrict__ c8,
                                               const uint32_t* __restrict__ cp, uint32_t cka, uint32_t ckb,
                                               uint32_t& m1a, uint32_t& m2a, uint32_t& m1b, uint32_t& m2b) {
    const uint32_t ka2 = cka + 0x00020002u, kb2 = ckb + 0x00020002u;
#pragma unroll
    for (int d2 = 0; d2 < OTF_PART; d2 += 4) {
        const uint4 k4 = *(const uint4*)(key + d2), v8 = *(const uint4*)(c8 + d2), vp = *(const uint4*)(cp + d2);
        const uint32_t ka[4] = {k4.x, k4.y, k4.z, k4.w}, c8a[4] = {v8.x, v8.y, v8.z, v8.w}, cpa[4] = {vp.x, vp.y, vp.z, vp.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t ta = pk_sub(ka2, ka[i]), tb = pk_sub(kb2, ka[i]);
            const bool nra = pk_min(ta, 0x00040004u) == ta, eqa = ta == 0x00020002u;
            const bool nrb = pk_min(tb, 0x00040004u) == tb, eqb = tb == 0x00020002u;
            m1a = eqa ? c8a[i] : m1a;                                         // last match wins
            uint32_t sa = nra ? cpa[i] : OTF_BIG;
            sa = eqa ? OTF_BIG : sa;
            m2a = min(m2a, sa);
            m1b = eqb ? c8a[i] : m1b;
            uint32_t sb = nrb ? cpa[i] : OTF_BIG;
            sb = eqb ? OTF_BIG : sb;
            m2b = min(m2b, sb);
        }
    }
}

__device__ __forceinline__ void otf_match_part_exact(const int32_t* qx, const int32_t* qy, const uint32_t* c8, const uint32_t* cp,
                                                     int mvx, int mvy, uint32_t& m1, uint32_t& m2) {
    for (int d2 = 0; d2 < OTF_PART; d2++) {
        if (mvx == qx[d2] && mvy == qy[d2]) m1 = c8[d2];
        else if (near2(mvx, qx[d2]) && near2(mvy, qy[d2])) m2 = min(m2, cp[d2]);
    }
}

__global__ __launch_bounds__(896) void otf_pipe_kernel(OtfArgs a) {
    __shared__ Cand sC[3][OTF_D];                 // candidates of pixels n-1, n, n+1 (slot = pixel % 3)
    __shared__ Cand sL1[2][OTF_E];                // L1 double buffer (:197)
    __shared__ __attribute__((aligned(16))) uint32_t sKey[4][OTF_D], sC8[4][OTF_D], sCp[4][OTF_D];   // staged predecessor entries per path
    __shared__ int32_t sX[4][OTF_D], sY[4][OTF_D];                                                   // the same, unpacked (exact form)
    __shared__ uint32_t sPart[4][2][128][2];      // [path][part - 1][candidate][last match, near minimum]
    __shared__ int sUnsafe;
    __shared__ Cand sOut[2][4][OTF_E];            // new entries per path of pixels n-1, n (slot = pixel & 1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int path = wave / 3, part = wave - 3 * path;           // waves 0..11
    const bool is_path = wave < 12;
    const int W = a.W, H = a.H, NP = W * H;
    const size_t f = blockIdx.x;
    const uint8_t* __restrict__ If = a.I1 + f * (size_t)NP;
    const uint32_t* __restrict__ cen1 = a.cen1 + f * (size_t)NP;
    const uint32_t* __restrict__ cen2 = a.cen2 + f * (size_t)NP;
    const int32_t* __restrict__ rnd = a.rnd + f * (size_t)NP * 8;
    const size_t rowE = (size_t)W * OTF_E;
    Cand* Lrow = a.Lrow + f * 6 * rowE;           // [L2,L3,L4][2][W][OTF_E]
    for (int i = tid; i < 2 * OTF_E; i += 896) { Cand z = {0, 0, 0}; sL1[i / OTF_E][i % OTF_E] = z; }   // :204
    if (tid == 0) sUnsafe = a.exact;
    __syncthreads();

    // ---- path waves: operands of pixel t fetched one step ahead
    const int dxw = path == 1 ? -1 : (path == 3 ? 1 : 0);
    auto starts = [&](int x, int y) -> bool {
        return path == 0 ? x == 0 : (path == 1 ? (x == 0 || y == 0) : (path == 2 ? y == 0 : (y == 0 || x == W - 1)));
    };
    Cand ent = {0, 0, 0}, top = {0, 0, 0};
    int pm = 0, ppre = 0, pcur = 0;
    auto fetch_path = [&](int t, Cand& E, Cand& T, int& M, int& PP, int& PC) {
        const int ty = t / W, tx = t - ty * W;
        PC = If[t];
        if (path == 0) {
            if (tx > 0) PP = If[t - 1];
            return;
        }
        const int rc = 1 ^ (ty & 1);
        const Cand* cur = Lrow + ((path - 1) * 2 + rc) * rowE + (size_t)tx * OTF_E;
        if (part == 0 && lane < 2) T = cur[OTF_D + lane];
        if (!starts(tx, ty)) {
            const Cand* psrc = Lrow + ((path - 1) * 2 + (rc ^ 1)) * rowE + (size_t)(tx + dxw) * OTF_E;
            if (lane < OTF_PART) E = psrc[OTF_PART * part + lane];
            M = psrc[OTF_D].cost;
            PP = If[t - W + dxw];
        }
    };
    // ---- cost lanes: candidate d of every pixel; its hint (:276-277, :148-149) fetched one step ahead
    const int cd = tid - 768;                                     // 0..127, candidates 0..107
    const int chi = cd / 9, ck = cd % 9, cl = chi / 3, ci = chi % 3;
    const int coffy = ck / 3 - 1, coffx = ck % 3 - 1;             // :153-154 offy outer
    const bool is_cost_lane = !is_path && cd < OTF_D;
    const bool hint_in_lds = cl == 0 && ci < 2;                   // L1's top entries live in LDS
    int hx = 0, hy = 0;
    auto fetch_hint = [&](int t, int& HX, int& HY) {
        if (hint_in_lds) return;
        if (ci < 2) {
            const int ty = t / W, tx = t - ty * W;
            const Cand* src = Lrow + ((cl - 1) * 2 + (1 ^ (ty & 1))) * rowE + (size_t)tx * OTF_E + OTF_D + ci;
            HX = src->mvx; HY = src->mvy;
        } else {
            HX = rnd[(size_t)t * 8 + 2 * cl] % 256 - 128;         // :148
            HY = rnd[(size_t)t * 8 + 2 * cl + 1] % 128 - 64;      // :149
        }
    };
    auto candidate = [&](int t, int mvx, int mvy) {               // (:122-186) clamp-border 5x5 mean of Hamming costs
        const int y = t / W, x = t - y * W;
        if (hint_in_lds) { const Cand h = sL1[1 ^ (t & 1)][OTF_D + ci]; mvx = h.mvx; mvy = h.mvy; }
        // all 50 census words requested before the first is used (left alone the compiler waits after every pair)
        uint32_t w1[25], w2[25];
#pragma unroll
        for (int ay = -2; ay <= 2; ay++) {
            const int y1 = clampi(y + ay, 0, H - 1);
            const int y2 = clampi((coffy + y1) + mvy, 0, H - 1);
#pragma unroll
            for (int ax = -2; ax <= 2; ax++) {
                const int x1 = clampi(x + ax, 0, W - 1);
                const int x2 = clampi((coffx + x1) + mvx, 0, W - 1);
                w1[5 * (ay + 2) + ax + 2] = cen1[(uint32_t)(W * y1 + x1)];
                w2[5 * (ay + 2) + ax + 2] = cen2[(uint32_t)(W * y2 + x2)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < 25; i++) sum += __popc(w1[i] ^ w2[i]);
        Cand c;
        c.cost = f64_to_i32_x86(__dadd_rn(__ddiv_rn(__dmul_rn(1.0, (double)sum), 25.0), 0.5));   // :177
        c.mvx = mvx + coffx; c.mvy = mvy + coffy;
        sC[t % 3][cd] = c;
        if (!(c.mvx > -0x3FF0 && c.mvx < 0x3FF0 && c.mvy > -0x3FF0 && c.mvy < 0x3FF0)) sUnsafe = 1;
    };
    auto pack = [](int mvx, int mvy) -> uint32_t { return ((uint32_t)(mvx + 0x4000) << 16) | ((uint32_t)(mvy + 0x4000) & 0xFFFFu); };

    // prologue: candidates of pixel 0, operands of pixel 0, hints of pixel 1
    if (is_path) {
        fetch_path(0, ent, top, pm, ppre, pcur);
    } else if (is_cost_lane) {
        fetch_hint(0, hx, hy);
        candidate(0, hx, hy);
        if (NP > 1) fetch_hint(1, hx, hy);
    }
    __syncthreads();

    for (int n = 0; n <= NP; n++) {
        const bool live = is_path && n < NP;
        const int y = n / W, x = n - y * W;
        const int l1cur = 1 ^ (n & 1), l1pre = l1cur ^ 1, rowcur = 1 ^ (y & 1);
        const bool start = live && starts(x, y);
        const bool has1 = lane + 64 < OTF_D;
        Cand nent = ent, ntop = top;
        int nm = pm, npp = ppre, npc = pcur;
        Cand ca = {0, 0, 0}, cb = {0, 0, 0};
        uint32_t m1a = OTF_NOMATCH, m2a = OTF_BIG, m1b = OTF_NOMATCH, m2b = OTF_BIG;
        Cand* out = sOut[n & 1][path & 3];
        // ================= phase 1: partial matchers | candidate costs of pixel n+1
        if (live) {
            if (n + 1 < NP) fetch_path(n + 1, nent, ntop, nm, npp, npc);         // consumed after the matcher
            const Cand* Cc = sC[n % 3];
            // current content of the slot being overwritten: its top-N mvs survive a start copy
            if (part == 0 && lane < 2) out[OTF_D + lane] = path == 0 ? sL1[l1cur][OTF_D + lane] : top;
            if (start) {
                if (part == 0) {
                    __builtin_amdgcn_wave_barrier();
                    for (int d = lane; d < OTF_D; d += 64) out[d] = Cc[d];       // :280,284,290,294,297,304
                    if (lane == 0) out[OTF_D].cost = 0;
                }
            } else {
                const int d0 = OTF_PART * part;
                if (lane < OTF_PART) {
                    const Cand e = path == 0 ? sL1[l1pre][d0 + lane] : ent;
                    sX[path][d0 + lane] = e.mvx; sY[path][d0 + lane] = e.mvy;
                    sKey[path][d0 + lane] = pack(e.mvx, e.mvy);
                    sC8[path][d0 + lane] = (uint32_t)e.cost & 0xFF;
                    sCp[path][d0 + lane] = (uint32_t)(e.cost + a.P1) & 0xFF;
                }
                ca = Cc[lane]; cb = Cc[has1 ? lane + 64 : lane];
                __builtin_amdgcn_wave_barrier();
                if (sUnsafe == 0) {
                    otf_match_part(sKey[path] + d0, sC8[path] + d0, sCp[path] + d0, pack(ca.mvx, ca.mvy), pack(cb.mvx, cb.mvy),
                                   m1a, m2a, m1b, m2b);
                } else {
                    otf_match_part_exact(sX[path] + d0, sY[path] + d0, sC8[path] + d0, sCp[path] + d0, ca.mvx, ca.mvy, m1a, m2a);
                    otf_match_part_exact(sX[path] + d0, sY[path] + d0, sC8[path] + d0, sCp[path] + d0, cb.mvx, cb.mvy, m1b, m2b);
                }
                if (part > 0) {
                    sPart[path][part - 1][lane][0] = m1a; sPart[path][part - 1][lane][1] = m2a;
                    sPart[path][part - 1][lane + 64][0] = m1b; sPart[path][part - 1][lane + 64][1] = m2b;
                }
            }
        } else if (!is_path) {
            int nhx = hx, nhy = hy;
            if (is_cost_lane && n + 2 < NP) fetch_hint(n + 2, nhx, nhy);
            if (is_cost_lane && n + 1 < NP) candidate(n + 1, hx, hy);
            hx = nhx; hy = nhy;
        }
        __syncthreads();
        // ================= phase 2: fold the parts, finish the path step | WTA of pixel n-1
        if (live && part == 0) {
            if (!start) {
                const uint32_t m = (uint32_t)(path == 0 ? sL1[l1pre][OTF_D].cost : pm) & 0xFF;      // :53
                const int P2 = abs(pcur - ppre) > 50 ? a.P2 / 8 : a.P2;                             // :101-105 adaptive P2
                const uint32_t jump = (m + (uint32_t)P2) & 0xFF;
#pragma unroll
                for (int e = 0; e < 2; e++) {                                                       // later parts override an earlier match
                    const uint32_t pa = sPart[path][e][lane][0], pb = sPart[path][e][lane + 64][0];
                    m1a = pa != OTF_NOMATCH ? pa : m1a;
                    m1b = pb != OTF_NOMATCH ? pb : m1b;
                    m2a = min(m2a, sPart[path][e][lane][1]);
                    m2b = min(m2b, sPart[path][e][lane + 64][1]);
                }
                const uint32_t besta = min(jump, min(m1a == OTF_NOMATCH ? jump : m1a, m2a));
                const uint32_t bestb = min(jump, min(m1b == OTF_NOMATCH ? jump : m1b, m2b));
                Cand oa = ca, ob = cb;
                oa.cost = (ca.cost + (int)besta) - (int)m;
                ob.cost = (cb.cost + (int)bestb) - (int)m;
                out[lane] = oa;
                if (has1) out[lane + 64] = ob;
                otf_keep_best(out, oa.cost, ob.cost, has1, lane);
            }
            __builtin_amdgcn_wave_barrier();
            // write the new entries back into the path's buffer
            Cand* dst = path == 0 ? sL1[l1cur] : Lrow + ((path - 1) * 2 + rowcur) * rowE + (size_t)x * OTF_E;
            for (int d = lane; d < OTF_E; d += 64) dst[d] = out[d];
        } else if (wave == 13 && n >= 1) {
            // S = L1+L3+L2+L4 (:357-362) and WTA (:389-407) of pixel n-1
            const int t = n - 1;
            const Cand (*o)[OTF_E] = sOut[t & 1];
            // first minimum of the u32 sums in d order (a negative sum wraps and sorts last, as in the reference)
            const bool has1w = lane + 64 < OTF_D;
            const uint32_t s0 = (uint32_t)(o[0][lane].cost + o[2][lane].cost) + (uint32_t)(o[1][lane].cost + o[3][lane].cost);
            const int d1 = has1w ? lane + 64 : lane;
            const uint32_t s1 = (uint32_t)(o[0][d1].cost + o[2][d1].cost) + (uint32_t)(o[1][d1].cost + o[3][d1].cost);
            const uint32_t smin = wave_min_u32(min(s0, s1));
            const uint32_t idx = wave_min_u32(s0 == smin ? (uint32_t)lane : (s1 == smin ? (uint32_t)d1 : 0xFFFFFFFFu));
            if (lane == 0) {
                a.minC[f * NP + t] = smin;
                a.flow[f * 2 * (size_t)NP + t] = (double)sC[t % 3][idx].mvx;
                a.flow[f * 2 * (size_t)NP + NP + t] = (double)sC[t % 3][idx].mvy;
            }
        }
        ent = nent; top = ntop; pm = nm; ppre = npp; pcur = npc;
        __syncthreads();
    }
}

// =============================================================================================
// launchers
// =============================================================================================
void launch_ng_cost(hipStream_t st, const NgCostArgs& a, int frames) {
    static const bool by_hint = [] { const char* e = getenv("FSGM_NG_COST_HINT"); return !(e && e[0] == '0'); }();   // A/B switch
    if (by_hint && a.rX == 1 && a.rY == 1 && a.rAgg == 1) {
        const long long n = (long long)a.W * a.H * 9;
        const dim3 grid((unsigned)((n + 255) / 256), frames);
        if (a.K4 && a.flags) {                                // 4-byte entries, and the 12-byte list only if a key did not fit (gated on the device)
            hipLaunchKernelGGL(ng_cost_hint_kernel<true>, grid, dim3(256), 0, st, a);
            hipLaunchKernelGGL(ng_cost_hint_kernel<false>, grid, dim3(256), 0, st, a);
        } else {
            NgCostArgs b = a;
            b.flags = nullptr;
            hipLaunchKernelGGL(ng_cost_hint_kernel<false>, grid, dim3(256), 0, st, b);
        }
        return;
    }
    const long long n = (long long)a.W * a.H * 9 * (2 * a.rX + 1) * (2 * a.rY + 1);
    dim3 grid((unsigned)((n + 255) / 256), frames);
    hipLaunchKernelGGL(ng_cost_kernel, grid, dim3(256), 0, st, a);
}

void launch_ng_aggregate(hipStream_t st, NgAggArgs a, int frames) {
    // slots: 0 along x, 1 along y, 2/3 their point mirrors (pass 1)
    int acc = 0;
    for (int i = 0; i < 4; i++) {
        a.blk_begin[i] = acc;
        acc += (((i & 1) == 0 ? a.H : a.W) + 3) / 4;
    }
    a.blk_begin[4] = acc;
    a.role = NG_ROLE_ANY; a.with_compact = 0; a.compact_force = 0; a.compact_g = 0;
    { const char* e = getenv("FSGM_NG_DEDUPE"); if (e && atoi(e) == 0) { a.dd = nullptr; a.dk = nullptr; a.dbox = nullptr; a.ck = nullptr; a.cm = nullptr; } }   // A/B switch: stage every candidate
    if (a.D <= 128 && a.unsafe) {
        const int lpb = 256 / a.D, Dp = (a.D + 3) & ~3;
        // long lines first: with few frames their blocks decide when the launch ends
        const char* env = getenv("FSGM_NG_SPLIT");               // A/B switch: parts per matcher (0/1: one thread per (line, candidate))
        const int nparts = frames <= 2 ? (env ? atoi(env) : 2) : 1;   // 1242x375, 3-level pyramid: 9.64 / 7.34 / 7.98 / 7.66 ms with 1 / 2 / 3 / 4 parts
        const bool split = nparts >= 2 && nparts <= 4;
        const int ord_x[4] = {0, 2, 1, 3}, ord_y[4] = {1, 3, 0, 2};
        // Which kernel runs is settled on the device from what the dedupe kernel saw (ng_agg_not_mine): every candidate
        // is launched, the ones not favoured return at once.  A/B switches: FSGM_NG_COMPACT=0 / FSGM_NG_GRID=0 take a kernel out
        // of the set, FSGM_NG_GRID=1 makes the grid kernel the only one, FSGM_NG_COMPACT_G=16|32|64 fixes the compact kernel's
        // lanes a line (it still steps aside for lists it cannot hold).
        const char* cenv = getenv("FSGM_NG_COMPACT");
        const int compact_env = cenv && *cenv ? atoi(cenv) : -1;
        const char* genv = getenv("FSGM_NG_GRID");
        const int grid_env = genv && *genv ? atoi(genv) : -1;
        const bool can_stat = a.dd && a.dk && a.kstat;
        if (can_stat && a.ck && a.cm && a.L4 && compact_env != 0 && grid_env != 1) {
            acc = 0;
            for (int i = 0; i < 4; i++) {
                const int sl = a.W >= a.H ? ord_x[i] : ord_y[i];
                a.slot_of_c[i] = sl;
                a.blk_begin_c[i] = acc;
                acc += (((sl & 1) == 0 ? a.H : a.W) + 3) / 4;
            }
            a.blk_begin_c[4] = acc;
            a.with_compact = 1;
            a.role = NG_ROLE_COMPACT;
            // one launch per lanes-a-line class (16 / 32 / 64 for lists up to that long); the dedupe kernel's flags pick one
            const char* gforce = getenv("FSGM_NG_COMPACT_G");     // 16 / 32 / 64: that class only, whatever the lists look like (tests)
            const int g_only = gforce && *gforce ? atoi(gforce) : 0;
            a.compact_force = g_only == 16 || g_only == 32 || g_only == 64;
            for (int G = 16; G <= 64; G *= 2) {
                if (a.compact_force && G != g_only) continue;
                const int lpb = 4 * (64 / G);                    // lines a workgroup
                acc = 0;
                for (int i = 0; i < 4; i++) {
                    a.blk_begin_c[i] = acc;
                    acc += (((a.slot_of_c[i] & 1) == 0 ? a.H : a.W) + lpb - 1) / lpb;
                }
                a.blk_begin_c[4] = acc;
                a.compact_g = G;
                if (G == 16)      hipLaunchKernelGGL(ng_agg_compact_kernel<16>, dim3(acc, frames), dim3(256), 0, st, a);
                else if (G == 32) hipLaunchKernelGGL(ng_agg_compact_kernel<32>, dim3(acc, frames), dim3(256), 0, st, a);
                else              hipLaunchKernelGGL(ng_agg_compact_kernel<64>, dim3(acc, frames), dim3(256), 0, st, a);
            }
        }
        acc = 0;
        for (int i = 0; i < 4; i++) {
            const int sl = split ? (a.W >= a.H ? ord_x[i] : ord_y[i]) : i;
            a.slot_of[i] = sl;
            a.blk_begin[i] = acc;
            acc += (((sl & 1) == 0 ? a.H : a.W) + lpb - 1) / lpb;
        }
        a.blk_begin[4] = acc;
        // The grid form of the matcher costs the same at any list length, the list form grows with it: lists of a few
        // entries (nearly constant hint maps) are faster walked, anything richer is faster looked up.
        const bool with_grid = can_stat && a.dbox && grid_env != 0 && (grid_env == 1 || !split);
        if (with_grid) {
            const size_t lds = ((size_t)lpb * 8 * Dp + (size_t)lpb * 6 * NG_GCELLS + lpb * 8 + 2 * lpb) * sizeof(uint32_t);
            a.role = grid_env == 1 ? NG_ROLE_ANY : NG_ROLE_GRID;
            hipLaunchKernelGGL(ng_agg_grid_kernel, dim3(acc, frames), dim3(256), lds, st, a);
            if (grid_env == 1) return;
        }
        a.role = with_grid ? NG_ROLE_LIST : (a.with_compact ? NG_ROLE_REST : NG_ROLE_ANY);
        if (split) {
            const size_t lds = ((size_t)lpb * (10 * Dp + 8) + (size_t)(nparts - 1) * lpb * Dp * 2) * sizeof(uint32_t);
            if (nparts == 2)      hipLaunchKernelGGL(ng_agg_split_kernel<2>, dim3(acc, frames), dim3(512), lds, st, a);
            else if (nparts == 3) hipLaunchKernelGGL(ng_agg_split_kernel<3>, dim3(acc, frames), dim3(768), lds, st, a);
            else                  hipLaunchKernelGGL(ng_agg_split_kernel<4>, dim3(acc, frames), dim3(1024), lds, st, a);
            return;
        }
        const size_t lds = ((size_t)lpb * 8 * Dp + lpb * 8) * sizeof(uint32_t);
        hipLaunchKernelGGL(ng_agg_lines_kernel, dim3(acc, frames), dim3(256), lds, st, a);
        return;
    }
    hipLaunchKernelGGL(ng_agg_kernel, dim3(acc, frames), dim3(256), 0, st, a);
}

void launch_ng_fill_repeats(hipStream_t st, uint32_t* S, const uint16_t* dd, const uint16_t* cm, int W, int H, int D, int frames) {
    const long long n = (long long)W * H * D * frames;
    hipLaunchKernelGGL(ng_fill_repeats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, dd, cm, n, D);
}

void launch_ng_dedupe(hipStream_t st, const Cand* C, uint16_t* dd, uint8_t* dk, uint32_t* dbox, uint32_t* kstat, uint32_t* ck, uint16_t* cm, int W, int H, int D, int frames,
                      const uint32_t* K4, const uint32_t* flags) {
    const int n = W * H * frames;                            // frames are contiguous in all arrays
    if (kstat) (void)hipMemsetAsync(kstat, 0, NG_KSTAT_WORDS * sizeof(uint32_t), st);
    { const char* e = getenv("FSGM_NG_GRID"); if (e && e[0] == '0') dbox = nullptr; }      // no grid kernel in the set: no boxes needed
    hipLaunchKernelGGL(ng_dedupe_kernel, dim3((n + 3) / 4), dim3(256), 0, st, C, (K4 && flags) ? K4 : nullptr, flags, dd, dk, dbox, kstat, ck, cm, n, D);
}

// ---- 4-byte entries: what happens between the dedupe kernel and the matchers (all decided on the device, nothing read back) ----
// kstat[258] := 1 when the compact matcher is the one that will run (the predicate of ng_agg_not_mine): S is not used then, and
// with every key in range neither is the 12-byte list
__global__ __launch_bounds__(256) void ng_decide_kernel(uint32_t* __restrict__ kstat, long long npix_all, int with_compact) {
    __shared__ uint32_t sSum;
    if (threadIdx.x == 0) sSum = 0;
    __syncthreads();
    uint32_t v = kstat[threadIdx.x];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += (uint32_t)__shfl_xor((int)v, s);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sSum, v);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long npix = (((unsigned long long)(npix_all + 3) / 4 + 15) / 16) * 4;      // the dedupe kernel's sample
        kstat[258] = (with_compact && (kstat[256] & 3u) == 0u && (unsigned long long)sSum < (unsigned long long)NG_COMPACT_MAX_K * npix) ? 1u : 0u;
    }
}
// the general matchers read 12-byte entries: made from the keys when the keys are all there is (in range, but the lists too long for the compact matcher)
__global__ __launch_bounds__(256) void ng_expand_kernel(const uint32_t* __restrict__ K4, Cand* __restrict__ C, const uint32_t* __restrict__ flags,
                                                        const uint32_t* __restrict__ kstat, long long n) {
    if (flags[1] != 0 || kstat[258] != 0) return;            // (a few thousand workgroups that loop: an early return costs nothing then)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) C[i] = ng_unkey4(K4[i]);
}
// S shares its memory with the keys: zeroed (calc_pyd_cost_sgm_ng.cpp:111) once they are no longer needed, and only when a matcher will add to it
__global__ __launch_bounds__(256) void ng_zero_s_kernel(uint4* __restrict__ S4, const uint32_t* __restrict__ kstat, long long n16, int tail) {
    if (kstat[258] != 0) return;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    for (long long i = i0; i < n16; i += (long long)gridDim.x * 256) S4[i] = make_uint4(0, 0, 0, 0);
    if (i0 == 0) { uint32_t* t = (uint32_t*)(S4 + n16); for (int k = 0; k < tail; k++) t[k] = 0; }
}

// The bounding box of a pixel's motion vectors (what the grid matcher stages by): the dedupe kernel's own job when it is asked for
// one, but a quarter of its instructions (four wave reductions) -- with 4-byte entries it is left to this kernel, which only
// runs when the compact matcher does not
__global__ __launch_bounds__(256) void ng_dbox_kernel(const uint32_t* __restrict__ K4, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ kstat,
                                                      uint32_t* __restrict__ dbox, int NPtot, int D) {
    if (flags[1] != 0 || kstat[258] != 0) return;            // (flags[1] != 0: the dedupe kernel read 12-byte entries and made the boxes itself)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = blockIdx.x * 4 + wave; p < NPtot; p += gridDim.x * 4) {
        const uint32_t* k = K4 + (size_t)p * D;
        const Cand f0 = ng_unkey4(k[min(lane, D - 1)]), f1 = ng_unkey4(k[min(lane + 64, D - 1)]);
        const uint32_t xl = wave_min_u32((uint32_t)(min(f0.mvx, f1.mvx) + 0x4000)), xh = 0x8000u - wave_min_u32((uint32_t)(0x4000 - max(f0.mvx, f1.mvx)));
        const uint32_t yl = wave_min_u32((uint32_t)(min(f0.mvy, f1.mvy) + 0x4000)), yh = 0x8000u - wave_min_u32((uint32_t)(0x4000 - max(f0.mvy, f1.mvy)));
        if (lane == 0) dbox[p] = (xh - xl < (uint32_t)NG_GB && yh - yl < (uint32_t)NG_GB) ? ((xl << 16) | yl) : NG_BOX_WIDE;
    }
}

bool ng_compact_possible(const NgAggArgs& a) {
    const char* cenv = getenv("FSGM_NG_COMPACT");
    const int compact_env = cenv && *cenv ? atoi(cenv) : -1;
    const char* genv = getenv("FSGM_NG_GRID");
    const int grid_env = genv && *genv ? atoi(genv) : -1;
    const char* denv = getenv("FSGM_NG_DEDUPE");
    if (denv && atoi(denv) == 0) return false;
    // (the kernel walks a frame with 32-bit byte offsets: entries x 4 bytes and pixels x 512 bytes of L4 must stay below 4 GB)
    if ((long long)a.W * a.H * a.D >= (1LL << 30) || (long long)a.W * a.H >= (1LL << 23)) return false;
    return a.D <= 128 && a.unsafe && a.dd && a.dk && a.kstat && a.ck && a.cm && a.L4 && compact_env != 0 && grid_env != 1;
}

void launch_ng_prepare_matchers(hipStream_t st, const NgAggArgs& a, const uint32_t* K4, Cand* C, const uint32_t* flags, int frames) {
    const long long npix = (long long)a.W * a.H * frames, n = npix * a.D;
    hipLaunchKernelGGL(ng_decide_kernel, dim3(1), dim3(256), 0, st, a.kstat, npix, ng_compact_possible(a) ? 1 : 0);
    hipLaunchKernelGGL(ng_expand_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, st, K4, C, flags, (const uint32_t*)a.kstat, n);
    if (a.dbox) hipLaunchKernelGGL(ng_dbox_kernel, dim3((unsigned)std::min<long long>((npix + 3) / 4, 4096)), dim3(256), 0, st, K4, flags, (const uint32_t*)a.kstat,
                                   const_cast<uint32_t*>(a.dbox), (int)npix, a.D);
    hipLaunchKernelGGL(ng_zero_s_kernel, dim3((unsigned)std::min<long long>((n / 4 + 256) / 256, 4096)), dim3(256), 0, st, (uint4*)a.S, (const uint32_t*)a.kstat, n / 4, (int)(n % 4));
}

__global__ __launch_bounds__(256) void ng_l4_to_s_kernel(uint32_t* __restrict__ S, const int16_t* __restrict__ L4, const uint16_t* __restrict__ cm,
                                                         const uint8_t* __restrict__ dk, const uint32_t* __restrict__ kstat, long long npix, int D) {
    if (kstat[257] == 0) return;                              // another matcher ran: S is complete already
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long p = i >> 6;
    const int e = (int)(i & 63);
    if (p >= npix || e >= (int)dk[p]) return;
    const int16_t* Lp = L4 + p * NG_L4_PER_PIXEL;
    S[p * D + (cm[p * D + e] >> 8)] = (uint32_t)((int)Lp[e] + (int)Lp[64 + e] + (int)Lp[128 + e] + (int)Lp[192 + e]);
}

void launch_ng_l4_to_s(hipStream_t st, uint32_t* S, const int16_t* L4, const uint16_t* cm, const uint8_t* dk, const uint32_t* kstat, int W, int H, int D, int frames) {
    const long long npix = (long long)W * H * frames;
    hipLaunchKernelGGL(ng_l4_to_s_kernel, dim3((unsigned)((npix * 64 + 255) / 256)), dim3(256), 0, st, S, L4, cm, dk, kstat, npix, D);
}

void launch_ng_wta(hipStream_t st, const NgWtaArgs& a, int frames) {
    hipLaunchKernelGGL(ng_wta_kernel, dim3((a.W * a.H + 15) / 16, frames), dim3(256), 0, st, a);
}

void launch_ng_subpixel(hipStream_t st, const NgSubpixArgs& a, int frames) {
    hipLaunchKernelGGL(ng_subpixel_kernel, dim3((a.W * a.H + 255) / 256, frames), dim3(256), 0, st, a);
}

void launch_otf(hipStream_t st, const OtfArgs& a, int frames) {
    // the pipelined form fetches a pixel's operands one step ahead: they must have been written two steps back
    if (a.W >= 4) hipLaunchKernelGGL(otf_pipe_kernel, dim3(frames), dim3(896), 0, st, a);
    else          hipLaunchKernelGGL(otf_kernel, dim3(frames), dim3(256), 0, st, a);
}

}  // namespace fsgm
