/*
 * fsgm_oracle_pyd.cpp -- CPU restatement of calc_pyd_cost_sgm.cpp (reference: /root/reference,
 * cited per function as file:line).
 *
 * TEST INFRASTRUCTURE ONLY (see fsgm_oracle.h).  PARITY UNPINNED: calc_pyd_cost_sgm.cpp includes
 * MATLAB's mex.h, absent from this image, so it cannot be built here without a stand-in header,
 * and the reference ships no golden vectors for it.
 *
 * Organised per path direction, like fsgm_oracle_epi.cpp: each direction is an independent
 * recurrence; S is the u32 sum of all of them.
 */
#include "fsgm_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

inline int32_t f64_to_i32(double v) {               /* x86-64 cvttsd2si */
    if (v > -2147483649.0 && v < 2147483648.0) return (int32_t)v;
    return INT32_MIN;
}
inline uint8_t f64_to_u8(double v) { return (uint8_t)(uint32_t)f64_to_i32(v); }
inline uint8_t u8(int v) { return (uint8_t)v; }

/* calc_pyd_cost_sgm.cpp:34-89.  (dx,dy) = hint(cur) - hint(path predecessor). */
void step_2d(uint8_t* L, const uint8_t* Lpre, const uint8_t* C, double dx, double dy,
             int Sx, int Sy, int P1, int P2) {
    const int D = Sx * Sy;
    const uint8_t m = Lpre[D];
    const uint8_t jump = u8(m + P2);                              /* :50-53 */
    uint8_t lowest = 255;                                         /* :40 */
    for (int sx = 0; sx < Sx; sx++)
        for (int sy = 0; sy < Sy; sy++) {
            const int ypre = f64_to_i32(sy + dy + 0.5);           /* :46  (sy + dy) + 0.5, truncation */
            const int xpre = f64_to_i32(sx + dx + 0.5);           /* :47 */
            uint8_t min1 = jump, min2 = jump;
            if (xpre >= 0 && xpre < Sx && ypre >= 0 && ypre < Sy) min1 = Lpre[xpre * Sy + ypre];   /* :56-59 */
            for (int k = -2; k <= 2; k++)                         /* :61-76 */
                for (int mm = -2; mm <= 2; mm++) {
                    if (mm == 0 && k == 0) continue;
                    /* int overflow of xpre+mm cannot happen for in-range tests that matter: an
                       xpre of INT_MIN wraps in the reference (UB); keep it out of range here */
                    const long long ty = (long long)ypre + k, tx = (long long)xpre + mm;
                    if (tx >= 0 && tx < Sx && ty >= 0 && ty < Sy) {
                        const uint8_t t = u8(Lpre[tx * Sy + ty] + P1);
                        if (t < min2) min2 = t;
                    }
                }
            uint8_t best = jump;
            if (min1 < best) best = min1;
            if (min2 < best) best = min2;
            const int d = sx * Sy + sy;
            L[d] = u8((C[d] + best) - m);                         /* :83 */
            if (L[d] < lowest) lowest = L[d];
        }
    L[D] = lowest;                                                /* :88 */
}

inline int adaptive_P2(int P2, int cur, int pre) {                /* :91-95 */
    return abs(cur - pre) > 50 ? P2 / 8 : P2;
}

/* one direction r=(rx,ry); predecessor p-r (calc_pyd_cost_sgm.cpp:167-277) */
void aggregate_dir_2d(uint32_t* S, const uint8_t* I1, const uint8_t* C, int W, int H,
                      const double* mvx, const double* mvy, int mvW,
                      int Sx, int Sy, int rx, int ry, int P1, int P2, int adaptive, uint32_t weight) {
    const int D = Sx * Sy, E = D + 1;
    std::vector<uint8_t> bufA((size_t)W * E), bufB((size_t)W * E);
    uint8_t* prev = bufA.data();
    uint8_t* cur = bufB.data();
    const int ys = ry >= 0 ? 1 : -1, y0 = ry >= 0 ? 0 : H - 1;
    const int xs = rx >= 0 ? 1 : -1, x0 = rx >= 0 ? 0 : W - 1;
    for (int yi = 0, y = y0; yi < H; yi++, y += ys) {
        for (int xi = 0, x = x0; xi < W; xi++, x += xs) {
            const int px = x - rx, py = y - ry;
            const bool inside = px >= 0 && px < W && py >= 0 && py < H;
            uint8_t* L = cur + (size_t)x * E;
            const uint8_t* c = C + ((size_t)y * W + x) * D;
            if (!inside) {
                memcpy(L, c, D);                                  /* :181-182 etc. */
                L[D] = 0;
            } else {
                const double dx = mvx[(size_t)y * mvW + x] - mvx[(size_t)py * mvW + px];   /* :213,226,240,253 */
                const double dy = mvy[(size_t)y * mvW + x] - mvy[(size_t)py * mvW + px];
                const int p2 = adaptive ? adaptive_P2(P2, I1[(size_t)W * y + x], I1[(size_t)W * py + px]) : P2;
                const uint8_t* Lp = (ry == 0 ? cur : prev) + (size_t)px * E;
                step_2d(L, Lp, c, dx, dy, Sx, Sy, P1, p2);
            }
            uint32_t* s = S + ((size_t)y * W + x) * D;
            for (int d = 0; d < D; d++) s[d] += weight * L[d];
        }
        uint8_t* t = prev; prev = cur; cur = t;
    }
}

}  // namespace

extern "C" {

/* calc_pyd_cost_sgm.cpp:374-437 */
void fsgm_oracle_pyd_cost(uint8_t* C, const uint32_t* cen1, const uint32_t* cen2, int W, int H,
                          const double* preMv, int mvW, int mvH, int rAgg, int rX, int rY) {
    const double* pMvx = preMv;
    const double* pMvy = preMv + (size_t)mvW * mvH;               /* :379-380 */
    const int winPixels = (2 * rAgg + 1) * (2 * rAgg + 1);
    const int D = (2 * rX + 1) * (2 * rY + 1);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint8_t* out = C + ((size_t)y * W + x) * D;
            const double mvx = pMvx[(size_t)mvW * y + x], mvy = pMvy[(size_t)mvW * y + x];   /* :388-389 own stride */
            int d = 0;
            for (int offx = -rX; offx <= rX; offx++)              /* :392 x offset is the slow index */
                for (int offy = -rY; offy <= rY; offy++) {
                    unsigned sum = 0;
                    for (int ay = -rAgg; ay <= rAgg; ay++)
                        for (int ax = -rAgg; ax <= rAgg; ax++) {
                            const int y1 = y + ay, x1 = x + ax;
                            if (y1 < 0 || y1 > H - 1 || x1 < 0 || x1 > W - 1) { sum += 5; continue; }   /* :405-408 */
                            const int y2 = f64_to_i32(1.0 * (offy + y1) + mvy + 0.5);                  /* :415 */
                            const int x2 = f64_to_i32(1.0 * (offx + x1) + mvx + 0.5);                  /* :416 */
                            if (y2 < 0 || y2 > H - 1 || x2 < 0 || x2 > W - 1) { sum += 5; continue; }   /* :418-421 */
                            sum += (unsigned)__builtin_popcount(cen1[(size_t)W * y1 + x1] ^ cen2[(size_t)W * y2 + x2]);
                        }
                    out[d++] = f64_to_u8((1.0 * sum / winPixels) + 0.5);   /* :431 */
                }
        }
}

/* calc_pyd_cost_sgm.cpp:114-296 */
void fsgm_oracle_pyd_aggregate(uint32_t* S, const uint8_t* I1, const uint8_t* C, int W, int H,
                               const double* preMv, int mvW, int mvH, int Sx, int Sy,
                               int P1, int P2, int diagonal, int totalPass, int adaptiveP2) {
    const int D = Sx * Sy;
    memset(S, 0, sizeof(uint32_t) * (size_t)W * H * D);           /* :126 */
    const double* mvx = preMv;
    const double* mvy = preMv + (size_t)mvW * mvH;                /* :128-129 */
    static const int dirs[4][2] = {{1, 0}, {0, 1}, {1, 1}, {-1, 1}};
    const int nd = diagonal ? 4 : 2;
    /* :142-151: pass 0 runs forward; every later pass runs with the mirrored start/step (they are
     * set once at pass==1 and never reset), so passes 1..totalPass-1 are identical. */
    for (int pass = 0; pass < totalPass && pass < 2; pass++) {
        const int sgn = pass == 0 ? 1 : -1;
        const uint32_t weight = pass == 0 ? 1u : (uint32_t)(totalPass - 1);
        for (int k = 0; k < nd; k++)
            aggregate_dir_2d(S, I1, C, W, H, mvx, mvy, mvW, Sx, Sy, sgn * dirs[k][0], sgn * dirs[k][1],
                             P1, P2, adaptiveP2, weight);
    }
}

/* calc_pyd_cost_sgm.cpp:298-364.  mvSub: f64 [2][H][W], plane 0 = x; left untouched (zero from
 * mxCreateNumericArray) when subpixel == 0. */
void fsgm_oracle_pyd_wta(uint32_t* bestD, uint32_t* minC, double* mvSub, const uint32_t* S,
                         int W, int H, int Sx, int Sy, int subpixel) {
    const int D = Sx * Sy;
    const size_t NP = (size_t)W * H;
    for (size_t p = 0; p < NP; p++) {
        const uint32_t* s = S + p * D;
        uint32_t lo = s[0], idx = 0;
        for (int d = 1; d < D; d++)
            if (s[d] < lo) { lo = s[d]; idx = d; }
        minC[p] = lo;
        bestD[p] = idx;
        if (!subpixel) continue;
        const double c0 = (double)s[idx];
        const int dx = idx / Sy, dy = idx % Sy;                    /* :333-334 */
        if (dy > 0 && dy < Sy - 1) {
            const double cl = (double)s[idx - 1], cr = (double)s[idx + 1];
            mvSub[NP + p] = cr < cl ? (cr - cl) / (c0 - cl) / 2.0 : (cr - cl) / (c0 - cr) / 2.0;   /* :340-343 */
        } else mvSub[NP + p] = 0;
        if (dx > 0 && dx < Sx - 1) {
            const double cl = (double)s[idx - Sy], cr = (double)s[idx + Sy];
            mvSub[p] = cr < cl ? (cr - cl) / (c0 - cl) / 2.0 : (cr - cl) / (c0 - cr) / 2.0;        /* :353-356 */
        } else mvSub[p] = 0;
    }
}

/* calc_pyd_cost_sgm.cpp:439-510 */
void fsgm_oracle_calc_pyd_cost_sgm(uint32_t* bestD, uint32_t* minC, double* mvSub,
                                   const uint8_t* I1, const uint8_t* I2, int W, int H,
                                   const double* preMv, int mvW, int mvH,
                                   int rX, int rY, int rAgg, int subpixel, int P1, int P2,
                                   int diagonal, int totalPass, int adaptiveP2,
                                   uint8_t* C_out, uint32_t* S_out) {
    const size_t NP = (size_t)W * H;
    const int Sx = 2 * rX + 1, Sy = 2 * rY + 1, D = Sx * Sy;
    std::vector<uint32_t> cen1(NP), cen2(NP), Sbuf;
    std::vector<uint8_t> Cbuf;
    fsgm_oracle_census(I1, cen1.data(), W, H, 2);                 /* :485-486 */
    fsgm_oracle_census(I2, cen2.data(), W, H, 2);
    uint8_t* C = C_out;
    uint32_t* S = S_out;
    if (!C) { Cbuf.resize(NP * D); C = Cbuf.data(); }
    if (!S) { Sbuf.resize(NP * D); S = Sbuf.data(); }
    fsgm_oracle_pyd_cost(C, cen1.data(), cen2.data(), W, H, preMv, mvW, mvH, rAgg, rX, rY);          /* :498 */
    fsgm_oracle_pyd_aggregate(S, I1, C, W, H, preMv, mvW, mvH, Sx, Sy, P1, P2, diagonal, totalPass, adaptiveP2);
    memset(mvSub, 0, sizeof(double) * 2 * NP);                    /* :476 zero-initialised output */
    fsgm_oracle_pyd_wta(bestD, minC, mvSub, S, W, H, Sx, Sy, subpixel);
}

}  // extern "C"
