// pyd_kernels.hip -- gfx950 kernels for the pyramidal 2-D variant
// (reference: calc_pyd_cost_sgm.cpp; citations per kernel).
//
// Cost volume C and per-path costs L_r: u8 [H][W][D], D = Sx*Sy, candidate index
// d = sx*Sy + sy (x offset is the slow index, calc_pyd_cost_sgm.cpp:392-393).
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
    const int win = (2 * r + 1) * (2 * r + 1);
    const double v = __dadd_rn(__ddiv_rn(__dmul_rn(1.0, (double)sum), (double)win), 0.5);         // :431
    a.C[f * (size_t)NP * D + (size_t)p * D + d] = (uint8_t)(uint32_t)f64_to_i32_x86(v);
}

// =============================================================================================
// 2-D path aggregation  (calc_pyd_cost_sgm.cpp:34-89 sgm_step, :114-296 sgm2d).
// One wave per path line; the previous pixel's D+1 path costs live in LDS; lanes stride over the
// candidates and read the 5x5 neighbourhood around the hint-shifted centre from LDS.  Exact u8
// semantics for any P1/P2 (each neighbour + P1 is narrowed before the min, centre excluded).
// =============================================================================================
__global__ __launch_bounds__(256) void pyd_agg_kernel(PydAggArgs a) {
    __shared__ uint8_t sL[4][2][FSGM_PYD_MAX_D + 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int slot = 0;
#pragma unroll
    for (int i = 1; i < 8; i++)
        if (i < a.ndirs && (int)blockIdx.x >= a.blk_begin[i]) slot = i;
    const int code = a.dir_code[slot];
    const int base = code & 3;
    const bool mirror = (code & 4) != 0;
    const int W = a.W, H = a.H, Sx = a.Sx, Sy = a.Sy, D = Sx * Sy;
    const int NP = W * H;
    const int nlines = base == 0 ? H : W;
    const int len = base == 0 ? W : H;
    const int line = ((int)blockIdx.x - a.blk_begin[slot]) * 4 + wave;
    if (line >= nlines) return;                              // wave-uniform
    const size_t f = blockIdx.y;
    const uint8_t* __restrict__ Cf = a.C + f * (size_t)NP * D;
    const uint8_t* __restrict__ If = a.I1 + f * (size_t)NP;
    uint8_t* __restrict__ Lf = a.L + (f * a.ndirs + slot) * (size_t)NP * D;
    const double* mvxp = a.mv + f * 2 * (size_t)a.mvW * a.mvH;
    const double* mvyp = mvxp + (size_t)a.mvW * a.mvH;
    // path direction in pass-0 coordinates; the predecessor of p is p - r
    const int rx = base == 1 ? 0 : (base == 3 ? -1 : 1), ry = base == 0 ? 0 : 1;
    int x = base == 0 ? 0 : line, y = base == 0 ? line : 0;  // pass-0 frame coordinates of step 0
    uint8_t* pre = sL[wave][0];
    uint8_t* cur = sL[wave][1];
    uint32_t m = 0;
    for (int t = 0; t < len; t++) {
        const bool start = (t == 0) || (base == 2 && x == 0) || (base == 3 && x == W - 1);
        const int ax = mirror ? W - 1 - x : x, ay = mirror ? H - 1 - y : y;      // actual pixel
        const int px = mirror ? ax + rx : ax - rx, py = mirror ? ay + ry : ay - ry;   // actual predecessor
        const size_t off = ((size_t)ay * W + ax) * D;
        uint32_t lo = 255;
        if (start) {
            for (int d = lane; d < D; d += 64) {
                const uint32_t v = Cf[off + d];
                cur[d] = (uint8_t)v;
                Lf[off + d] = (uint8_t)v;
            }
            m = 0;                                                               // :182 stored minimum 0
        } else {
            const double dx = __dsub_rn(mvxp[(size_t)ay * a.mvW + ax], mvxp[(size_t)py * a.mvW + px]);   // :213 etc.
            const double dy = __dsub_rn(mvyp[(size_t)ay * a.mvW + ax], mvyp[(size_t)py * a.mvW + px]);
            int P2 = a.P2;
            if (a.adaptive) {                                                    // :91-95
                const int dI = abs((int)If[(size_t)W * ay + ax] - (int)If[(size_t)W * py + px]);
                P2 = dI > 50 ? a.P2 / 8 : a.P2;
            }
            const uint32_t jump = (m + (uint32_t)P2) & 0xFF;                     // :50-53
            for (int d = lane; d < D; d += 64) {
                const int sx = d / Sy, sy = d - sx * Sy;
                const int ypre = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)sy, dy), 0.5));   // :46
                const int xpre = f64_to_i32_x86(__dadd_rn(__dadd_rn((double)sx, dx), 0.5));   // :47
                uint32_t best = jump;
                if (xpre >= 0 && xpre < Sx && ypre >= 0 && ypre < Sy) best = min(best, (uint32_t)pre[xpre * Sy + ypre]);   // :56-59
                // neighbours within +-2 of the shifted centre, centre excluded (:61-76)
                if (xpre >= -2 && xpre < Sx + 2 && ypre >= -2 && ypre < Sy + 2) {
#pragma unroll
                    for (int mm = -2; mm <= 2; mm++) {
                        const int tx = xpre + mm;
                        if (tx < 0 || tx >= Sx) continue;
#pragma unroll
                        for (int k = -2; k <= 2; k++) {
                            const int ty = ypre + k;
                            if ((mm == 0 && k == 0) || ty < 0 || ty >= Sy) continue;
                            best = min(best, ((uint32_t)pre[tx * Sy + ty] + (uint32_t)a.P1) & 0xFF);
                        }
                    }
                }
                const uint32_t v = ((uint32_t)Cf[off + d] + best - m) & 0xFF;    // :83
                cur[d] = (uint8_t)v;
                Lf[off + d] = (uint8_t)v;
                lo = min(lo, v);
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, s));
            m = lo;                                                              // :88
        }
        __builtin_amdgcn_wave_barrier();
        uint8_t* tmp = pre; pre = cur; cur = tmp;
        // advance in the pass-0 frame
        if (base == 0) x++;
        else {
            y++;
            if (base == 2) { x++; if (x == W) x = 0; }
            if (base == 3) { x--; if (x < 0) x = W - 1; }
        }
    }
}

// =============================================================================================
// WTA + y/x parabola  (calc_pyd_cost_sgm.cpp:298-364).  One wave per pixel.
// =============================================================================================
__device__ __forceinline__ uint32_t pyd_sum_at(const PydWtaArgs& a, const uint8_t* Lf, size_t vol, size_t idx) {
    uint32_t s = 0;
    for (int r = 0; r < a.ndirs; r++) s += a.weight[r] * (uint32_t)Lf[r * vol + idx];
    return s;
}

__device__ __forceinline__ double parabola(double cl, double c0, double cr) {
    return cr < cl ? __ddiv_rn(__ddiv_rn(__dsub_rn(cr, cl), __dsub_rn(c0, cl)), 2.0)
                   : __ddiv_rn(__ddiv_rn(__dsub_rn(cr, cl), __dsub_rn(c0, cr)), 2.0);
}

__global__ __launch_bounds__(256) void pyd_wta_kernel(PydWtaArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NP = a.W * a.H, Sy = a.Sy, Sx = a.Sx, D = Sx * Sy;
    const int p = blockIdx.x * 4 + wave;
    if (p >= NP) return;
    const size_t f = blockIdx.y;
    const size_t vol = (size_t)NP * D;
    const uint8_t* __restrict__ Lf = a.L + f * a.ndirs * vol;
    uint32_t lo = 0xFFFFFFFFu, idx = 0xFFFFFFFFu;
    for (int d = lane; d < D; d += 64) {
        const uint32_t s = pyd_sum_at(a, Lf, vol, (size_t)p * D + d);
        if (a.S) a.S[f * vol + (size_t)p * D + d] = s;
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
            const size_t b = (size_t)p * D + gidx;
            if (dy > 0 && dy < Sy - 1)
                suby = parabola((double)pyd_sum_at(a, Lf, vol, b - 1), c0, (double)pyd_sum_at(a, Lf, vol, b + 1));
            if (dx > 0 && dx < Sx - 1)
                subx = parabola((double)pyd_sum_at(a, Lf, vol, b - Sy), c0, (double)pyd_sum_at(a, Lf, vol, b + Sy));
        }
        a.mvSub[f * 2 * (size_t)NP + p] = subx;              // zero when subpixel is off (:476 zero-init output)
        a.mvSub[f * 2 * (size_t)NP + NP + p] = suby;
    }
}

// =============================================================================================
// launchers
// =============================================================================================
void launch_pyd_cost(hipStream_t st, const PydCostArgs& a, int frames) {
    const long long n = (long long)a.W * a.H * (2 * a.rX + 1) * (2 * a.rY + 1);
    dim3 grid((unsigned)((n + 255) / 256), frames);
    hipLaunchKernelGGL(pyd_cost_kernel, grid, dim3(256), 0, st, a);
}

int plan_pyd_dirs(PydAggArgs& a, int diagonal, int totalPass, uint32_t weight[8]) {
    // pass 0: along x, along y, (+1,+1), (-1,+1); later passes: their point mirrors, all identical
    // (calc_pyd_cost_sgm.cpp:142-151 sets the mirrored start/step once at pass==1)
    static const int fwd[4] = {0, 1, 2, 3};
    const int nd = diagonal ? 4 : 2;
    int n = 0, acc = 0;
    auto add = [&](int code, uint32_t w) {
        a.dir_code[n] = code;
        a.blk_begin[n] = acc;
        weight[n] = w;
        const int nlines = (code & 3) == 0 ? a.H : a.W;
        acc += (nlines + 3) / 4;
        n++;
    };
    if (totalPass >= 1) for (int k = 0; k < nd; k++) add(fwd[k], 1u);
    if (totalPass >= 2) for (int k = 0; k < nd; k++) add(fwd[k] | 4, (uint32_t)(totalPass - 1));
    a.ndirs = n;
    for (int i = n; i <= 8; i++) a.blk_begin[i] = acc;
    for (int i = n; i < 8; i++) { a.dir_code[i] = 0; weight[i] = 0; }
    return n;
}

void launch_pyd_aggregate(hipStream_t st, const PydAggArgs& a, int frames) {
    if (a.ndirs == 0) return;
    dim3 grid(a.blk_begin[8], frames);
    hipLaunchKernelGGL(pyd_agg_kernel, grid, dim3(256), 0, st, a);
}

void launch_pyd_wta(hipStream_t st, const PydWtaArgs& a, int frames) {
    dim3 grid((a.W * a.H + 3) / 4, frames);
    hipLaunchKernelGGL(pyd_wta_kernel, grid, dim3(256), 0, st, a);
}

}  // namespace fsgm
