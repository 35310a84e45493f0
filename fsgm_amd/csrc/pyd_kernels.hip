// pyd_kernels.hip -- gfx950 kernels for the pyramidal 2-D variant
// (reference: calc_pyd_cost_sgm.cpp; citations per kernel).
//
// Cost volume C and per-path costs L_r: u8 [H][W][PS]; candidate d = sx*Sy + sy (x offset is the
// slow index, calc_pyd_cost_sgm.cpp:392-393) sits at byte sx*RS + sy of its pixel.  RS = Sy,
// PS = D is the reference's own order; the padded "rows" layout (pyd_kernels.h) is what the
// row-packed kernels of pyd_rows.hip use.  The kernels in this file accept either.
#include "pyd_kernels.h"
#include "fsgm_device.h"

namespace fsgm {

// =============================================================================================
// 2-D window cost  (calc_pyd_cost_sgm.cpp:374-437).  One thread = one (pixel, candidate).
// The hint is the CENTRE pixel's (own stride mvW, :388-389); taps outside either image add the
// constant 5 (USE_CONST_COST, :32,405-421); sample index = (int)(1.0*(off+p1) + mv + 0.5),
// C truncation (values in (-1,0) become 0 and count as inside).
// =============================================================================================
__global__ __launch_bounds__(256) void pyd_cost_kernel(PydCostArgs a) {
    const int W = a.W, H = a.H;
    const int NP = W * H;
    const int Sy = 2 * a.rY + 1, D = (2 * a.rX + 1) * Sy;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)NP * D) return;
    const int p = (int)(gid / D), d = (int)(gid - (long long)p * D);
    const int y = p / W, x = p - y * W;
    const int offx = d / Sy - a.rX, offy = d % Sy - a.rY;
    const size_t f = blockIdx.y;
    const double* mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
    const double* mvyp = mvxp + (size_t)a.mvW * a.mvH;
    const double mvx = mvxp[(size_t)a.mvW * y + x], mvy = mvyp[(size_t)a.mvW * y + x];
    const uint32_t* cen1 = a.cen1 + f * (size_t)NP;
    const uint32_t* cen2 = a.cen2 + f * (size_t)NP;
    const int r = a.rAgg;
    uint32_t sum = 0;
    if (r <= 2) {
        // the reference's window (aggHalfWinSize 2, pyramidal_sgm.m:17): x2 depends on ax only and
        // y2 on ay only, so convert each once (10 conversions instead of 50); -1 marks a tap that
        // falls outside either image
        int xs1[5], xs2[5], ys1[5], ys2[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int t = k - 2;
            const int x1 = x + t, y1 = y + t;
            const int x2 = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)(offx + x1), mvx), 0.5));   // :416
            const int y2 = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)(offy + y1), mvy), 0.5));   // :415
            const bool in_win = t >= -r && t <= r;
            const bool xok = in_win && x1 >= 0 && x1 <= W - 1 && x2 >= 0 && x2 <= W - 1;
            const bool yok = in_win && y1 >= 0 && y1 <= H - 1 && y2 >= 0 && y2 <= H - 1;
            xs1[k] = xok ? x1 : -1; xs2[k] = x2;
            ys1[k] = yok ? y1 : -1; ys2[k] = y2;
        }
#pragma unroll
        for (int ky = 0; ky < 5; ky++) {
            if (ky - 2 < -r || ky - 2 > r) continue;
#pragma unroll
            for (int kx = 0; kx < 5; kx++) {
                if (kx - 2 < -r || kx - 2 > r) continue;
                if (ys1[ky] >= 0 && xs1[kx] >= 0)
                    sum += __popc(cen1[(size_t)W * ys1[ky] + xs1[kx]] ^ cen2[(size_t)W * ys2[ky] + xs2[kx]]);
                else
                    sum += 5;                                                                      // :406,:419
            }
        }
    } else {
        for (int ay = -r; ay <= r; ay++) {
            const int y1 = y + ay;
            const int y2 = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)(offy + y1), mvy), 0.5));   // :415
            const bool yok = y1 >= 0 && y1 <= H - 1 && y2 >= 0 && y2 <= H - 1;
            for (int ax = -r; ax <= r; ax++) {
                const int x1 = x + ax;
                const int x2 = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)(offx + x1), mvx), 0.5));   // :416
                if (yok && x1 >= 0 && x1 <= W - 1 && x2 >= 0 && x2 <= W - 1)
                    sum += __popc(cen1[(size_t)W * y1 + x1] ^ cen2[(size_t)W * y2 + x2]);
                else
                    sum += 5;                                                                          // :406,:419
            }
        }
    }
    const int win = (2 * r + 1) * (2 * r + 1);
    const double v = __dadd_rn(__ddiv_rn(__dmul_rn(1.0, (double)sum), (double)win), 0.5);         // :431
    a.C[(f * (size_t)NP + p) * a.PS + (d / Sy) * a.RS + d % Sy] = (uint8_t)(uint32_t)f64_to_i32_x86(v);
}

// =============================================================================================
// 2-D window cost, patch form (same result as pyd_cost_kernel).  One 128-thread workgroup per
// pixel: the sample column depends on offx+ax only and the sample row on offy+ay only
// (calc_pyd_cost_sgm.cpp:415-416), so the (Sx+2r) x (Sy+2r) patch of census codes of image 2 that
// all candidates and taps of this pixel touch is fetched once into LDS (225 loads instead of
// 2 x 3025 at 11x11 / 5x5), together with the (2r+1)^2 codes of image 1.  0xFFFFFFFF marks a tap
// that falls outside an image (a census code never has bit 0 set, common.cpp:21).
// =============================================================================================
constexpr uint32_t PYD_OUTSIDE = 0xFFFFFFFFu;
__global__ __launch_bounds__(128) void pyd_cost_patch_kernel(PydCostArgs a) {
    extern __shared__ uint32_t sPatch[];
    const int W = a.W, H = a.H, r = a.rAgg;
    const int NP = W * H;
    const int Sx = 2 * a.rX + 1, Sy = 2 * a.rY + 1, D = Sx * Sy;
    const int PX = Sx + 2 * r, PY = Sy + 2 * r, AW = 2 * r + 1;
    uint32_t* const patch = sPatch;                          // [PY][PX] census of image 2 at the sampled positions
    uint32_t* const c1s = patch + PX * PY;                   // [AW][AW]  census of image 1 around the pixel
    int* const x2tab = (int*)(c1s + AW * AW);                // [PX], [PY]: sampled column / row or -1
    int* const y2tab = x2tab + PX;
    const int p = blockIdx.x;
    const int y = p / W, x = p - y * W;
    const size_t f = blockIdx.y;
    const double* mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
    const double* mvyp = mvxp + (size_t)a.mvW * a.mvH;
    const double mvx = mvxp[(size_t)a.mvW * y + x], mvy = mvyp[(size_t)a.mvW * y + x];   // :388-389
    const uint32_t* cen1 = a.cen1 + f * (size_t)NP;
    const uint32_t* cen2 = a.cen2 + f * (size_t)NP;
    const int tid = threadIdx.x;
    for (int k = tid; k < PX + PY; k += 128) {
        if (k < PX) {                                        // k = (offx + rX) + (ax + r)  ->  offx + x1 = x + k - rX - r
            const int x2 = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)(x + k - a.rX - r), mvx), 0.5));   // :416
            x2tab[k] = (x2 >= 0 && x2 <= W - 1) ? x2 : -1;
        } else {
            const int kk = k - PX;
            const int y2 = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)(y + kk - a.rY - r), mvy), 0.5));  // :415
            y2tab[kk] = (y2 >= 0 && y2 <= H - 1) ? y2 : -1;
        }
    }
    for (int k = tid; k < AW * AW; k += 128) {
        const int y1 = y + k / AW - r, x1 = x + k % AW - r;
        c1s[k] = (y1 >= 0 && y1 <= H - 1 && x1 >= 0 && x1 <= W - 1) ? cen1[(size_t)W * y1 + x1] : PYD_OUTSIDE;   // :405
    }
    __syncthreads();
    for (int k = tid; k < PX * PY; k += 128) {
        const int ky = k / PX, kx = k - ky * PX;
        const int y2 = y2tab[ky], x2 = x2tab[kx];
        patch[k] = (y2 >= 0 && x2 >= 0) ? cen2[(size_t)W * y2 + x2] : PYD_OUTSIDE;                   // :418
    }
    __syncthreads();
    const int win = AW * AW;
    for (int d = tid; d < D; d += 128) {
        const int ox = d / Sy, oy = d - ox * Sy;             // offx + rX, offy + rY
        uint32_t sum = 0;
        for (int ay = 0; ay < AW; ay++)
            for (int ax = 0; ax < AW; ax++) {
                const uint32_t u = c1s[ay * AW + ax], v = patch[(oy + ay) * PX + ox + ax];
                sum += (u == PYD_OUTSIDE || v == PYD_OUTSIDE) ? 5u : (uint32_t)__popc(u ^ v);      // :406,:419,:427
            }
        const double vv = __dadd_rn(__ddiv_rn(__dmul_rn(1.0, (double)sum), (double)win), 0.5);     // :431
        a.C[(f * (size_t)NP + p) * a.PS + ox * a.RS + oy] = (uint8_t)(uint32_t)f64_to_i32_x86(vv);
    }
}

// =============================================================================================
// 2-D path aggregation  (calc_pyd_cost_sgm.cpp:34-89 sgm_step, :114-296 sgm2d).
// One wave per path line.  The previous pixel's path costs live in LDS as a (Sx+2*PADW) x
// (Sy+2*PADW) grid of u16 with an "absent" ring around the search window, so the 5x5
// neighbourhood around the hint-shifted centre needs no bounds tests; lanes stride over the
// candidates.  The hint shift is separable (xpre depends on sx only, ypre on sy only,
// calc_pyd_cost_sgm.cpp:46-47), so it is tabulated once per step by Sx+Sy lanes.  The next step's
// costs, hint delta and adaptive-P2 decision are fetched while the current step computes.
// WRAP=false needs 0<=P1,P2 and max C + P2 + max(P1,P2) <= 255 (no u8 narrowing changes a value:
// then including the centre cell in the "+P1" minimum is harmless and an absent cell is just a
// large number); WRAP=true narrows every neighbour + P1 to u8 first and excludes the centre,
// exactly like the reference, for any P1/P2.
// =============================================================================================
constexpr int PYD_PADW = 5;             // clamp(shifted centre) +- 2 stays inside the padded grid
constexpr int PYD_MAXS = 64;            // max search-window side supported by the tables

template <bool WRAP, int NCMAX>                            // NCMAX >= ceil(Sx*Sy / 64) candidates per lane
__global__ __launch_bounds__(256) void pyd_agg_kernel(PydAggArgs a) {
    extern __shared__ uint32_t sDynPyd[];   // u32 cells: aligned ds_read2_b32 for the neighbourhood rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int slot = 0;
#pragma unroll
    for (int i = 1; i < 8; i++)
        if (i < a.ndirs && (int)blockIdx.x >= a.blk_begin[i]) slot = i;
    const int code = a.dir_code[slot];
    const int base = code & 3;
    const bool mirror = (code & 4) != 0;
    const int W = a.W, H = a.H, Sx = a.Sx, Sy = a.Sy, D = Sx * Sy;
    const int GY = Sy + 2 * PYD_PADW, GN = (Sx + 2 * PYD_PADW) * GY;     // padded grid
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    const int line = ((int)blockIdx.x - a.blk_begin[slot]) * 4 + wave;
    // per-wave LDS: two padded grids + the two shift tables
    uint32_t* const wbase = sDynPyd + (size_t)wave * (2 * GN + 2 * PYD_MAXS);
    uint32_t* pre = wbase;
    uint32_t* cur = wbase + GN;
    int32_t* const xtab = (int32_t*)(wbase + 2 * GN);
    int32_t* const ytab = xtab + PYD_MAXS;
    if (line >= nlines) return;                              // wave-uniform
    const uint32_t ABSENT = 0x7F00;
    for (int i = lane; i < 2 * GN; i += 64) wbase[i] = ABSENT;
    const size_t f = blockIdx.y;
    const int RS = a.RS, PS = a.PS;
    const uint8_t* __restrict__ Cf = a.C + f * (size_t)NP * PS;
    const uint8_t* __restrict__ If = a.I1 + f * (size_t)NP;
    uint8_t* __restrict__ Lf = a.L + (f * a.ndirs + slot) * (size_t)NP * PS;
    const double* __restrict__ mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
    const double* __restrict__ mvyp = mvxp + (size_t)a.mvW * a.mvH;
    // path direction in pass-0 coordinates; the predecessor of p is p - r
    const int rx = base == 1 ? 0 : (base == 3 ? -1 : 1), ry = base == 0 ? 0 : 1;
    // this lane's candidates: grid cell of (sx, sy) and linear index d
    auto cell = [&](int sx, int sy) { return (sx + PYD_PADW) * GY + sy + PYD_PADW; };

    // cursor in the pass-0 frame + what is fetched per step
    int x = base == 0 ? 0 : line, y = base == 0 ? line : 0;
    auto actual = [&](int cx, int cy, int& ax, int& ay) { ax = mirror ? W - 1 - cx : cx; ay = mirror ? H - 1 - cy : cy; };
    auto is_start = [&](int t, int cx) { return (t == 0) || (base == 2 && cx == 0) || (base == 3 && cx == W - 1); };
    auto advance = [&](int& cx, int& cy) {
        if (base == 0) cx++;
        else {
            cy++;
            if (base == 2) { cx++; if (cx == W) cx = 0; }
            if (base == 3) { cx--; if (cx < 0) cx = W - 1; }
        }
    };
    struct Fetch { uint32_t c[NCMAX]; double dx, dy; int P2; };
    // branch-free (loads stay in flight across the step): at a path start the predecessor may lie
    // outside the image, its coordinates are clamped and the values are not used
    auto fetch = [&](int cx, int cy, Fetch& o) {
        int ax, ay;
        actual(cx, cy, ax, ay);
        const size_t off = ((size_t)ay * W + ax) * PS;
#pragma unroll
        for (int i = 0; i < NCMAX; i++) {
            const int d = min(lane + 64 * i, D - 1);
            o.c[i] = Cf[off + (d / Sy) * RS + d % Sy];
        }
        const int px = clampi(mirror ? ax + rx : ax - rx, 0, W - 1), py = clampi(mirror ? ay + ry : ay - ry, 0, H - 1);
        o.dx = __dsub_rn(mvxp[(size_t)ay * a.mvW + ax], mvxp[(size_t)py * a.mvW + px]);   // :213 etc.
        o.dy = __dsub_rn(mvyp[(size_t)ay * a.mvW + ax], mvyp[(size_t)py * a.mvW + px]);
        const int dI = abs((int)If[(size_t)W * ay + ax] - (int)If[(size_t)W * py + px]);
        o.P2 = (a.adaptive && dI > 50) ? a.P2 / 8 : a.P2;                            // :91-95
    };

    Fetch nxt;
    fetch(x, y, nxt);
    int xn = x, yn = y;
    advance(xn, yn);
    uint32_t m = 0;
    for (int t = 0; t < len; t++) {
        const Fetch now = nxt;
        {                                                    // in flight while this step computes
            const bool last = t + 1 >= len;
            fetch(last ? x : xn, last ? y : yn, nxt);
        }
        const bool start = is_start(t, x);
        int ax, ay;
        actual(x, y, ax, ay);
        const size_t off = ((size_t)ay * W + ax) * PS;
        uint32_t lo = 255;
        if (start) {
#pragma unroll
            for (int i = 0; i < NCMAX; i++) {
                const int d = lane + 64 * i;
                if (d < D) {
                    const int sx = d / Sy, sy = d - sx * Sy;
                    cur[cell(sx, sy)] = now.c[i];
                    Lf[off + sx * RS + sy] = (uint8_t)now.c[i];
                }
            }
            m = 0;                                                               // :182 stored minimum 0
        } else {
            // shift tables: xpre(sx), ypre(sy), clamped so that every tap stays in the padded grid
            if (lane < Sx) xtab[lane] = clampi(f64_to_i32_x86(__dadd_rn(__dadd_rn((double)lane, now.dx), 0.5)), -3, Sx + 2);   // :47
            if (lane < Sy) ytab[lane] = clampi(f64_to_i32_x86(__dadd_rn(__dadd_rn((double)lane, now.dy), 0.5)), -3, Sy + 2);   // :46
            __builtin_amdgcn_wave_barrier();
            const uint32_t jump = (m + (uint32_t)now.P2) & 0xFF;                 // :50-53
            const uint32_t P1 = WRAP ? (uint32_t)a.P1 & 0xFF : (uint32_t)a.P1;
#pragma unroll
            for (int i = 0; i < NCMAX; i++) {
                const int d = lane + 64 * i;
                if (d < D) {
                    const int sx = d / Sy, sy = d - sx * Sy;
                    const uint32_t* ctr = pre + cell(xtab[sx], ytab[sy]);
                    uint32_t best = min(jump, (uint32_t)ctr[0]);                 // :56-59 (absent centre = large)
                    uint32_t nb = 0xFFFFu;
#pragma unroll
                    for (int mm = -2; mm <= 2; mm++)
#pragma unroll
                        for (int k = -2; k <= 2; k++) {
                            if (WRAP && mm == 0 && k == 0) continue;             // :64
                            const uint32_t v = ctr[mm * GY + k];
                            // WRAP: u8(v + P1) for real cells (:73), absent cells stay >= 0x7F00
                            nb = min(nb, WRAP ? (((v + P1) & 0xFFu) | (v & 0xFF00u)) : v);
                        }
                    best = min(best, WRAP ? nb : nb + P1);
                    const uint32_t v = (now.c[i] + best - m) & 0xFF;             // :83
                    cur[cell(sx, sy)] = v;
                    Lf[off + sx * RS + sy] = (uint8_t)v;
                    lo = min(lo, v);
                }
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, s));
            m = lo;                                                              // :88
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t* tmp = pre; pre = cur; cur = tmp;
        x = xn; y = yn;
        advance(xn, yn);
    }
}

// =============================================================================================
// WTA + y/x parabola  (calc_pyd_cost_sgm.cpp:298-364).  One wave per pixel.
// =============================================================================================
__device__ __forceinline__ uint32_t pyd_sum_at(const PydWtaArgs& a, const uint8_t* Lf, size_t vol, size_t idx) {
    uint32_t v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = r < a.ndirs ? (uint32_t)Lf[r * vol + idx] : 0u;   // all loads in flight together
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) s += a.weight[r] * v[r];
    return s;
}


__global__ __launch_bounds__(256) void pyd_wta_kernel(PydWtaArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NP = a.W * a.H, Sy = a.Sy, Sx = a.Sx, D = Sx * Sy;
    const int p = blockIdx.x * 4 + wave;
    if (p >= NP) return;
    const size_t f = blockIdx.y;
    const size_t vol = (size_t)NP * a.PS;
    const uint8_t* __restrict__ Lf = a.L + f * a.ndirs * vol;
    auto at = [&](uint32_t d) { return (size_t)p * a.PS + (d / Sy) * a.RS + d % Sy; };
    uint32_t lo = 0xFFFFFFFFu, idx = 0xFFFFFFFFu;
    for (int d = lane; d < D; d += 64) {
        const uint32_t s = pyd_sum_at(a, Lf, vol, at(d));
        if (a.S) a.S[(f * NP + p) * D + d] = s;
        if (s < lo) { lo = s; idx = d; }                     // ascending d per lane: first minimum
    }
    uint32_t glo = lo;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) glo = min(glo, (uint32_t)__shfl_xor((int)glo, s));
    uint32_t gidx = lo == glo ? idx : 0xFFFFFFFFu;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) gidx = min(gidx, (uint32_t)__shfl_xor((int)gidx, s));
    if (lane == 0) {
        a.bestD[f * NP + p] = gidx;
        a.minC[f * NP + p] = glo;
        double subx = 0.0, suby = 0.0;
        if (a.subpixel) {
            const double c0 = (double)glo;
            const int dx = gidx / Sy, dy = gidx % Sy;                            // :333-334
            if (dy > 0 && dy < Sy - 1)
                suby = pyd_parabola((double)pyd_sum_at(a, Lf, vol, at(gidx - 1)), c0, (double)pyd_sum_at(a, Lf, vol, at(gidx + 1)));
            if (dx > 0 && dx < Sx - 1)
                subx = pyd_parabola((double)pyd_sum_at(a, Lf, vol, at(gidx - Sy)), c0, (double)pyd_sum_at(a, Lf, vol, at(gidx + Sy)));
        }
        a.mvSub[f * 2 * (size_t)NP + p] = subx;              // zero when subpixel is off (:476 zero-init output)
        a.mvSub[f * 2 * (size_t)NP + NP + p] = suby;
    }
}

// =============================================================================================
// launchers
// =============================================================================================
void launch_pyd_cost(hipStream_t st, const PydCostArgs& a, int frames) {
    if (pyd_rows_cost_ok(a)) { launch_pyd_rows_cost(st, a, frames); return; }
    const long long n = (long long)a.W * a.H * (2 * a.rX + 1) * (2 * a.rY + 1);
    const int PX = 2 * a.rX + 1 + 2 * a.rAgg, PY = 2 * a.rY + 1 + 2 * a.rAgg, AW = 2 * a.rAgg + 1;
    const size_t lds = (size_t)(PX * PY + AW * AW + PX + PY) * 4;
    if (lds <= 48 * 1024 && (long long)a.W * a.H < 2147483647LL) {      // the normal case: one workgroup per pixel
        hipLaunchKernelGGL(pyd_cost_patch_kernel, dim3(a.W * a.H, frames), dim3(128), lds, st, a);
        return;
    }
    dim3 grid((unsigned)((n + 255) / 256), frames);
    hipLaunchKernelGGL(pyd_cost_kernel, grid, dim3(256), 0, st, a);
}

int plan_pyd_dirs(PydAggArgs& a, int diagonal, int totalPass, uint32_t weight[8], int lines_per_block, bool wide_rows) {
    // pass 0: along x, along y, (+1,+1), (-1,+1); later passes: their point mirrors, all identical
    // (calc_pyd_cost_sgm.cpp:142-151 sets the mirrored start/step once at pass==1)
    static const int fwd[4] = {0, 1, 2, 3};
    const int nd = diagonal ? 4 : 2;
    int n = 0, acc = 0;
    a.wide_mask = 0;
    auto add = [&](int code, uint32_t w) {
        a.dir_code[n] = code;
        a.blk_begin[n] = acc;
        weight[n] = w;
        const int nlines = (code & 3) == 0 ? a.H : a.W;
        const bool wide = wide_rows && (code & 3) == 0;
        if (wide) a.wide_mask |= 1 << n;
        const int lpb = wide ? 4 : lines_per_block;
        acc += (nlines + lpb - 1) / lpb;
        n++;
    };
    // (the order of the slots is free -- volumes, descriptors and weights all go by slot.  The wide ones first: blocks are dispatched
    //  in order, so the long one-line-per-wave workgroups then start on CUs of their own and the packed ones fill in around them --
    //  0.69 -> 0.64 ms at 1242x375 against the order "pass 0, then pass 1"; handing the items out from a counter to 2-5 workgroups
    //  per CU instead, wide ones first, was no better: 0.66-0.72)
    if (wide_rows) {
        if (totalPass >= 1) add(fwd[0], 1u);
        if (totalPass >= 2) add(fwd[0] | 4, (uint32_t)(totalPass - 1));
    }
    if (totalPass >= 1) for (int k = wide_rows ? 1 : 0; k < nd; k++) add(fwd[k], 1u);
    if (totalPass >= 2) for (int k = wide_rows ? 1 : 0; k < nd; k++) add(fwd[k] | 4, (uint32_t)(totalPass - 1));
    a.ndirs = n;
    for (int i = n; i <= 8; i++) a.blk_begin[i] = acc;
    for (int i = n; i < 8; i++) { a.dir_code[i] = 0; weight[i] = 0; }
    return n;
}

void launch_pyd_aggregate(hipStream_t st, const PydAggArgs& a, int frames, bool wrap) {
    if (a.ndirs == 0) return;
    dim3 grid(a.blk_begin[8], frames);
    const int GN = (a.Sx + 2 * PYD_PADW) * (a.Sy + 2 * PYD_PADW);
    const size_t lds = (size_t)4 * (2 * GN + 2 * PYD_MAXS) * sizeof(uint32_t);
    const int nc = (a.Sx * a.Sy + 63) / 64;
#define FSGM_PYD_LAUNCH(NCM)                                                                        \
    do {                                                                                            \
        if (wrap) hipLaunchKernelGGL((pyd_agg_kernel<true, NCM>), grid, dim3(256), lds, st, a);    \
        else      hipLaunchKernelGGL((pyd_agg_kernel<false, NCM>), grid, dim3(256), lds, st, a);   \
    } while (0)
    if (nc <= 2) FSGM_PYD_LAUNCH(2);          // up to 11x11 (the reference's window)
    else if (nc <= 4) FSGM_PYD_LAUNCH(4);
    else FSGM_PYD_LAUNCH(16);
#undef FSGM_PYD_LAUNCH
}

void launch_pyd_wta(hipStream_t st, const PydWtaArgs& a, int frames) {
    if (pyd_rows_wta_ok(a)) { launch_pyd_rows_wta(st, a, frames); return; }
    dim3 grid((a.W * a.H + 3) / 4, frames);
    hipLaunchKernelGGL(pyd_wta_kernel, grid, dim3(256), 0, st, a);
}

}  // namespace fsgm
