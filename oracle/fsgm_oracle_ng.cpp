/*
 * fsgm_oracle_ng.cpp -- CPU restatement of the two neighbour-guided variants:
 *   calc_pyd_cost_sgm_ng.cpp (candidate list built from a hint map) and
 *   calc_cost_sgm_ng.cpp     (candidates taken on the fly from the path buffers + rand()).
 * Reference: /root/reference, cited per function as file:line.
 *
 * TEST INFRASTRUCTURE ONLY (see fsgm_oracle.h).  PARITY UNPINNED: both sources include MATLAB's
 * mex.h, absent from this image, and the reference ships no golden vectors.
 */
#include "fsgm_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

typedef fsgm_oracle_cand Cand;

inline int32_t f64_to_i32(double v) {               /* x86-64 cvttsd2si */
    if (v > -2147483649.0 && v < 2147483648.0) return (int32_t)v;
    return INT32_MIN;
}
inline uint8_t u8(int v) { return (uint8_t)v; }
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* |a-b| <= 2 on ints the way the reference writes it (abs of an int difference) */
inline bool near2(int a, int b) { const long long d = (long long)a - b; return d >= -2 && d <= 2; }

/* calc_pyd_cost_sgm_ng.cpp:39-78.  L/Lpre: D entries + 1 whose .cost is the running minimum. */
void step_ng(Cand* L, const Cand* Lpre, const Cand* C, int D, int P1, int P2) {
    uint8_t lowest = 255;                                          /* :45 */
    const uint8_t m = u8(Lpre[D].cost);                            /* :46 narrowed */
    const uint8_t jump = u8(m + P2);
    for (int d = 0; d < D; d++) {
        const int mvx = C[d].mvx, mvy = C[d].mvy;
        uint8_t min1 = jump, min2 = jump;
        for (int d2 = 0; d2 < D; d2++) {                           /* :57-67 */
            if (mvx == Lpre[d2].mvx && mvy == Lpre[d2].mvy) min1 = u8(Lpre[d2].cost);            /* last match wins */
            else if (near2(mvx, Lpre[d2].mvx) && near2(mvy, Lpre[d2].mvy)) {
                const uint8_t t = u8(Lpre[d2].cost + P1);
                if (t < min2) min2 = t;
            }
        }
        uint8_t best = jump;
        if (min1 < best) best = min1;
        if (min2 < best) best = min2;
        L[d].cost = (C[d].cost + best) - m;                        /* :71 kept as int */
        L[d].mvx = mvx;
        L[d].mvy = mvy;
        if (u8(L[d].cost) < lowest) lowest = u8(L[d].cost);        /* :74 narrowed */
    }
    L[D].cost = lowest;                                            /* :77 */
}

}  // namespace

extern "C" {

/* calc_pyd_cost_sgm_ng.cpp:370-446 */
void fsgm_oracle_ng_cost(Cand* C, const uint32_t* cen1, const uint32_t* cen2, int W, int H,
                         const double* preMv, int mvW, int mvH, int rAgg, int rX, int rY) {
    const double* pMvx = preMv;
    const double* pMvy = preMv + (size_t)mvW * mvH;
    const int winPixels = (2 * rAgg + 1) * (2 * rAgg + 1);
    const int D = 9 * (2 * rX + 1) * (2 * rY + 1);                 /* :494-497 hint radius 1 x 1 */
    const int step = 8;                                            /* :381 */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            Cand* out = C + ((size_t)y * W + x) * D;
            int d = 0;
            for (int dy = -step; dy <= step; dy += step)           /* :390 dy outer */
                for (int dx = -step; dx <= step; dx += step) {     /* :391 dx inner */
                    const int yn = clampi(y + dy, 0, mvH - 1), xn = clampi(x + dx, 0, mvW - 1);   /* :392-393 */
                    const double mvx = pMvx[(size_t)mvW * yn + xn], mvy = pMvy[(size_t)mvW * yn + xn];
                    for (int offx = -rX; offx <= rX; offx++)       /* :399 */
                        for (int offy = -rY; offy <= rY; offy++) { /* :400 */
                            unsigned sum = 0;
                            for (int ay = -rAgg; ay <= rAgg; ay++)
                                for (int ax = -rAgg; ax <= rAgg; ax++) {
                                    const int y1 = y + ay, x1 = x + ax;
                                    if (y1 < 0 || y1 > H - 1 || x1 < 0 || x1 > W - 1) { sum += 5; continue; }
                                    const int y2 = f64_to_i32((offy + y1) + mvy);      /* :417 no +0.5 */
                                    const int x2 = f64_to_i32((offx + x1) + mvx);      /* :418 */
                                    if (y2 < 0 || y2 > H - 1 || x2 < 0 || x2 > W - 1) { sum += 5; continue; }
                                    sum += (unsigned)__builtin_popcount(cen1[(size_t)W * y1 + x1] ^ cen2[(size_t)W * y2 + x2]);
                                }
                            out[d].cost = f64_to_i32((1.0 * sum / winPixels) + 0.5);   /* :432 */
                            out[d].mvx = f64_to_i32(mvx + offx);                        /* :433 */
                            out[d].mvy = f64_to_i32(mvy + offy);                        /* :434 */
                            d++;
                        }
                }
        }
}

/* calc_pyd_cost_sgm_ng.cpp:101-299: 2 passes x 2 paths (enableDiagnalPath=false :122), WTA
 * returning the winning candidate's motion vector. */
void fsgm_oracle_ng_aggregate_wta(uint32_t* minC, double* flow, uint32_t* S_out,
                                  const Cand* C, int W, int H, int D, int P1, int P2) {
    const size_t NP = (size_t)W * H;
    const int E = D + 1;
    std::vector<uint32_t> Sbuf;
    uint32_t* S = S_out;
    if (!S) { Sbuf.resize(NP * D); S = Sbuf.data(); }
    memset(S, 0, sizeof(uint32_t) * NP * D);
    static const int dirs[2][2] = {{1, 0}, {0, 1}};
    std::vector<Cand> bufA((size_t)W * E), bufB((size_t)W * E);
    for (int pass = 0; pass < 2; pass++)
        for (int k = 0; k < 2; k++) {
            const int sgn = pass == 0 ? 1 : -1;
            const int rx = sgn * dirs[k][0], ry = sgn * dirs[k][1];
            Cand* prev = bufA.data();
            Cand* cur = bufB.data();
            const int ys = ry >= 0 ? 1 : -1, y0 = ry >= 0 ? 0 : H - 1;
            const int xs = rx >= 0 ? 1 : -1, x0 = rx >= 0 ? 0 : W - 1;
            for (int yi = 0, y = y0; yi < H; yi++, y += ys) {
                for (int xi = 0, x = x0; xi < W; xi++, x += xs) {
                    const int px = x - rx, py = y - ry;
                    const bool inside = px >= 0 && px < W && py >= 0 && py < H;
                    Cand* L = cur + (size_t)x * E;
                    const Cand* c = C + ((size_t)y * W + x) * D;
                    if (!inside) {
                        memcpy(L, c, sizeof(Cand) * D);            /* :171-172,181-182 */
                        L[D].cost = 0;
                    } else {
                        const Cand* Lp = (ry == 0 ? cur : prev) + (size_t)px * E;
                        step_ng(L, Lp, c, D, P1, P2);              /* adpativeP2 = false :120 */
                    }
                    uint32_t* s = S + ((size_t)y * W + x) * D;
                    for (int d = 0; d < D; d++) s[d] += (uint32_t)L[d].cost;   /* :249 int added to unsigned */
                }
                Cand* t = prev; prev = cur; cur = t;
            }
        }
    for (size_t p = 0; p < NP; p++) {                              /* :281-299 */
        const uint32_t* s = S + p * D;
        uint32_t lo = s[0], idx = 0;
        for (int d = 1; d < D; d++)
            if (s[d] < lo) { lo = s[d]; idx = d; }
        minC[p] = lo;
        flow[p] = C[p * D + idx].mvx;
        flow[NP + p] = C[p * D + idx].mvy;
    }
}

/* calc_pyd_cost_sgm_ng.cpp:308-368 */
void fsgm_oracle_ng_subpixel(double* flow, const uint32_t* cen1, const uint32_t* cen2, int W, int H) {
    double* fx = flow;
    double* fy = flow + (size_t)W * H;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const uint32_t c1 = cen1[p];
            const int tx = f64_to_i32(fx[p] + x), ty = f64_to_i32(fy[p] + y);      /* :325-326 */
            if (!(tx > 1 && tx < W - 1 && ty > 1 && ty < H - 1)) continue;          /* :328 */
            const double c0 = __builtin_popcount(c1 ^ cen2[(size_t)ty * W + tx]);
            double cl = __builtin_popcount(c1 ^ cen2[(size_t)ty * W + tx - 1]);
            double cr = __builtin_popcount(c1 ^ cen2[(size_t)ty * W + tx + 1]);
            if (c0 >= cl || c0 >= cr) continue;                                     /* :337 skips the y part too */
            fx[p] += cr < cl ? (cr - cl) / (c0 - cl) / 2.0 : (cr - cl) / (c0 - cr) / 2.0;
            cl = __builtin_popcount(c1 ^ cen2[(size_t)(ty - 1) * W + tx]);
            cr = __builtin_popcount(c1 ^ cen2[(size_t)(ty + 1) * W + tx]);
            if (c0 >= cl || c0 >= cr) continue;                                     /* :354 */
            fy[p] += cr < cl ? (cr - cl) / (c0 - cl) / 2.0 : (cr - cl) / (c0 - cr) / 2.0;
        }
}

/* calc_pyd_cost_sgm_ng.cpp:448-523 */
void fsgm_oracle_calc_pyd_cost_sgm_ng(uint32_t* minC, double* flow,
                                      const uint8_t* I1, const uint8_t* I2, int W, int H,
                                      const double* preMv, int mvW, int mvH,
                                      double halfSearchWinSize, double aggSize, int subpixel,
                                      int P1, int P2, Cand* C_out, uint32_t* S_out) {
    const size_t NP = (size_t)W * H;
    const int r = f64_to_i32(halfSearchWinSize);                   /* :488-489 */
    const int rAgg = f64_to_i32(aggSize) / 2;                      /* :490 (int)aggSize/2 */
    const int D = 9 * (2 * r + 1) * (2 * r + 1);
    std::vector<uint32_t> cen1(NP), cen2(NP);
    fsgm_oracle_census(I1, cen1.data(), W, H, 2);
    fsgm_oracle_census(I2, cen2.data(), W, H, 2);
    std::vector<Cand> Cbuf;
    Cand* C = C_out;
    if (!C) { Cbuf.resize(NP * D); C = Cbuf.data(); }
    fsgm_oracle_ng_cost(C, cen1.data(), cen2.data(), W, H, preMv, mvW, mvH, rAgg, r, r);
    fsgm_oracle_ng_aggregate_wta(minC, flow, S_out, C, W, H, D, P1, P2);
    if (subpixel) fsgm_oracle_ng_subpixel(flow, cen1.data(), cen2.data(), W, H);   /* :516-517 */
}

/* ------------------------------------------------------------------------------------------
 * calc_cost_sgm_ng.cpp: on-the-fly neighbour-guided SGM.
 * Constants :5-11: M=1 random hint, N=2 best hints per path buffer, DX=DY=1, 4 directions,
 * aggHalfWin=2.  One forward pass with diagonals and adaptive P2 (:217-219).
 * ------------------------------------------------------------------------------------------ */
namespace {
const int NG_M = 1, NG_N = 2, NG_DIRS = 4;
const int NG_D = NG_DIRS * (NG_N + NG_M) * 9;                      /* :194 = 108 */
const int NG_E = NG_D + NG_N;                                      /* :196 entries per pixel */

/* calc_cost_sgm_ng.cpp:46-98 */
void step_otf(Cand* L, const Cand* Lpre, const Cand* C, int P1, int P2) {
    const int D = NG_D;
    const uint8_t m = u8(Lpre[D].cost);                            /* :53 */
    const uint8_t jump = u8(m + P2);
    for (int i = 0; i < NG_N; i++) L[D + i].cost = 255;            /* :54-55 (mv of these slots left as is) */
    for (int d = 0; d < D; d++) {
        const int mvx = C[d].mvx, mvy = C[d].mvy;
        uint8_t min1 = jump, min2 = jump;
        for (int d2 = 0; d2 < D; d2++) {
            if (mvx == Lpre[d2].mvx && mvy == Lpre[d2].mvy) min1 = u8(Lpre[d2].cost);
            else if (near2(mvx, Lpre[d2].mvx) && near2(mvy, Lpre[d2].mvy)) {
                const uint8_t t = u8(Lpre[d2].cost + P1);
                if (t < min2) min2 = t;
            }
        }
        uint8_t best = jump;
        if (min1 < best) best = min1;
        if (min2 < best) best = min2;
        L[d].cost = (C[d].cost + best) - m;                        /* :80 int */
        L[d].mvx = mvx;
        L[d].mvy = mvy;
        int j;                                                     /* :84-96 top-N insertion, int compare */
        for (j = 0; j < NG_N; j++)
            if (L[d].cost < L[D + j].cost) break;
        if (j < NG_N) {
            for (int i = NG_N - 1; i > j; i--) L[D + i] = L[D + i - 1];
            L[D + j] = L[d];
        }
    }
}

/* calc_cost_sgm_ng.cpp:122-186.  hint[l] points at the N best entries of path buffer l. */
void cost_from_hints(Cand* out, int x, int y, const uint32_t* cen1, const uint32_t* cen2, int W, int H,
                     const Cand* const hint[4], const int32_t*& rnd, const int32_t* rnd_end) {
    int cand = 0;
    for (int l = 0; l < NG_DIRS; l++)
        for (int i = 0; i < NG_N + NG_M; i++) {
            int mvx, mvy;
            if (i < NG_N) { mvx = hint[l][i].mvx; mvy = hint[l][i].mvy; }
            else {
                const int r0 = rnd < rnd_end ? *rnd++ : 0, r1 = rnd < rnd_end ? *rnd++ : 0;
                mvx = r0 % 256 - 128;                              /* :148 */
                mvy = r1 % 128 - 64;                               /* :149 */
            }
            for (int offy = -1; offy <= 1; offy++)                 /* :153 offy outer */
                for (int offx = -1; offx <= 1; offx++) {           /* :154 */
                    unsigned sum = 0;
                    for (int ay = -2; ay <= 2; ay++)
                        for (int ax = -2; ax <= 2; ax++) {
                            const int y1 = clampi(y + ay, 0, H - 1), x1 = clampi(x + ax, 0, W - 1);   /* :162-163 */
                            /* (offy+y1)+mvy is int arithmetic in the reference; mv is bounded by the
                               hint range so it cannot overflow */
                            const int y2 = clampi((offy + y1) + mvy, 0, H - 1);                      /* :167 */
                            const int x2 = clampi((offx + x1) + mvx, 0, W - 1);                      /* :168 */
                            sum += (unsigned)__builtin_popcount(cen1[(size_t)W * y1 + x1] ^ cen2[(size_t)W * y2 + x2]);
                        }
                    out[cand].cost = f64_to_i32(1.0 * sum / 25 + 0.5);                               /* :177 */
                    out[cand].mvx = mvx + offx;
                    out[cand].mvy = mvy + offy;
                    cand++;
                }
        }
}
}  // namespace

int64_t fsgm_oracle_sgm_ng_rand_draws(int W, int H) { return (int64_t)W * H * NG_DIRS * NG_M * 2; }

/* calc_cost_sgm_ng.cpp:188-419 + :484-526 (mexFunction ignores prhs[2..5]) */
void fsgm_oracle_calc_cost_sgm_ng(uint32_t* minC, double* flow,
                                  const uint8_t* I1, const uint8_t* I2, int W, int H,
                                  int P1, int P2, const int32_t* rand_stream, int64_t n_rand) {
    const size_t NP = (size_t)W * H;
    const int D = NG_D, E = NG_E;
    std::vector<uint32_t> cen1(NP), cen2(NP), S(NP * D, 0);
    fsgm_oracle_census(I1, cen1.data(), W, H, 2);                  /* :233-234 */
    fsgm_oracle_census(I2, cen2.data(), W, H, 2);
    /* The reference's double buffers, zero-initialised (:204-207).  Kept literally because the
     * hints are read from the buffer that is ABOUT to be overwritten (:276-277): L1 holds what was
     * written two pixels earlier in raster order, L2/L3/L4 what was written two rows earlier. */
    std::vector<Cand> L1(2 * (size_t)E), L2(2 * (size_t)W * E), L3(2 * (size_t)W * E), L4(2 * (size_t)W * E);
    std::vector<Cand> Cvol(NP * D);
    memset(L1.data(), 0, sizeof(Cand) * L1.size());
    memset(L2.data(), 0, sizeof(Cand) * L2.size());
    memset(L3.data(), 0, sizeof(Cand) * L3.size());
    memset(L4.data(), 0, sizeof(Cand) * L4.size());
    const int32_t* rnd = rand_stream;
    const int32_t* rnd_end = rand_stream + (rand_stream ? n_rand : 0);
    int l1pre = 0, l1cur = 1, rowpre = 0, rowcur = 1;
    const size_t rowE = (size_t)W * E;
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            Cand* pL1c = L1.data() + (size_t)l1cur * E;
            const Cand* pL1p = L1.data() + (size_t)l1pre * E;
            Cand* pL3c = L3.data() + rowcur * rowE + (size_t)x * E;
            const Cand* pL3p = L3.data() + rowpre * rowE + (size_t)x * E;
            Cand* pL2c = L2.data() + rowcur * rowE + (size_t)x * E;
            Cand* pL4c = L4.data() + rowcur * rowE + (size_t)x * E;
            Cand* c = Cvol.data() + ((size_t)y * W + x) * D;
            const Cand* hint[4] = {pL1c + D, pL2c + D, pL3c + D, pL4c + D};        /* :276-277 */
            cost_from_hints(c, x, y, cen1.data(), cen2.data(), W, H, hint, rnd, rnd_end);
            const int pc = I1[(size_t)W * y + x];
            if (x == 0) { memcpy(pL1c, c, sizeof(Cand) * D); pL1c[D].cost = 0; }   /* :279-281 */
            if (x == 0 || y == 0) { memcpy(pL2c, c, sizeof(Cand) * D); pL2c[D].cost = 0; }   /* :283-295 */
            if (y == 0) { memcpy(pL3c, c, sizeof(Cand) * D); pL3c[D].cost = 0; }   /* :290-291 */
            if (y == 0 || x == W - 1) { memcpy(pL4c, c, sizeof(Cand) * D); pL4c[D].cost = 0; }   /* :297-305 */
            if (x != 0) {                                                         /* :309-319 */
                const int pp = I1[(size_t)W * y + x - 1];
                step_otf(pL1c, pL1p, c, P1, abs(pc - pp) > 50 ? P2 / 8 : P2);
            }
            if (y != 0) {                                                         /* :322-330 */
                const int pp = I1[(size_t)W * (y - 1) + x];
                step_otf(pL3c, pL3p, c, P1, abs(pc - pp) > 50 ? P2 / 8 : P2);
            }
            if (x != 0 && y != 0) {                                               /* :333-342 */
                const int pp = I1[(size_t)W * (y - 1) + x - 1];
                step_otf(pL2c, L2.data() + rowpre * rowE + (size_t)(x - 1) * E, c, P1, abs(pc - pp) > 50 ? P2 / 8 : P2);
            }
            if (x != W - 1 && y != 0) {                                           /* :344-354 */
                const int pp = I1[(size_t)W * (y - 1) + x + 1];
                step_otf(pL4c, L4.data() + rowpre * rowE + (size_t)(x + 1) * E, c, P1, abs(pc - pp) > 50 ? P2 / 8 : P2);
            }
            uint32_t* s = S.data() + ((size_t)y * W + x) * D;
            for (int d = 0; d < D; d++)                                           /* :357-362 */
                s[d] += (uint32_t)(pL1c[d].cost + pL3c[d].cost) + (uint32_t)(pL2c[d].cost + pL4c[d].cost);
            const int t = l1pre; l1pre = l1cur; l1cur = t;                        /* :365-367 */
        }
        const int t = rowpre; rowpre = rowcur; rowcur = t;                        /* :371-384 */
    }
    for (size_t p = 0; p < NP; p++) {                                             /* :389-407 */
        const uint32_t* s = S.data() + p * D;
        uint32_t lo = s[0], idx = 0;
        for (int d = 1; d < D; d++)
            if (s[d] < lo) { lo = s[d]; idx = d; }
        minC[p] = lo;
        flow[p] = Cvol[p * D + idx].mvx;
        flow[NP + p] = Cvol[p * D + idx].mvy;
    }
}

}  // extern "C"
