/*
 * fsgm_oracle_epi.cpp -- CPU restatement of calc_cost_sgm.cpp + common.cpp (reference:
 * /root/reference, cited per function as file:line).
 *
 * TEST INFRASTRUCTURE ONLY (see fsgm_oracle.h).  PARITY UNPINNED except census(), which is
 * checked bit-for-bit against the reference's own common.cpp built into oracle/_ref/.
 *
 * The restatement is organised per path direction (one independent recurrence per direction,
 * summed into S) instead of the reference's four-paths-per-raster-pass loop; the two are the
 * same function because no path reads another path's state and S is a plain u32 sum.
 */
#include "fsgm_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

/* x86-64 gcc conversions (cvttsd2si): out-of-range / NaN give the "integer indefinite". */
inline int32_t f64_to_i32(double v) {
    if (v > -2147483649.0 && v < 2147483648.0) return (int32_t)v;
    return INT32_MIN;
}
inline int64_t f64_to_i64(double v) {
    if (v >= -9223372036854775808.0 && v < 9223372036854775808.0) return (int64_t)v;
    return INT64_MIN;
}
inline uint32_t f64_to_u32(double v) { return (uint32_t)(uint64_t)f64_to_i64(v); }
inline uint8_t  f64_to_u8(double v)  { return (uint8_t)(uint32_t)f64_to_i32(v); }

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }   /* common.h:12 */
inline uint8_t u8(int v) { return (uint8_t)v; }

/* calc_cost_sgm.cpp:33-66.  Lpre/L hold D costs + 1 trailing "minimum" entry. */
void step_1d(uint8_t* L, const uint8_t* Lpre, const uint8_t* C, int D, int P1, int P2) {
    const uint8_t m = Lpre[D];
    const uint8_t jump = u8(m + P2);                 /* :46,:51  LpreMin + P2 narrowed to u8 */
    uint8_t lowest = 255;                            /* :39 */
    for (int d = 0; d < D; d++) {
        uint8_t best = jump;
        if (Lpre[d] < best) best = Lpre[d];          /* :43,:55 */
        if (d > 0)     { uint8_t t = u8(Lpre[d - 1] + P1); if (t < best) best = t; }   /* :47 */
        if (d < D - 1) { uint8_t t = u8(Lpre[d + 1] + P1); if (t < best) best = t; }   /* :48 */
        L[d] = u8((C[d] + best) - m);                /* :60 */
        if (L[d] < lowest) lowest = L[d];            /* :61 */
    }
    L[D] = lowest;                                   /* :65 */
}

/* One path direction r=(dx,dy): L_r(p) = C(p) with min-entry 0 when p-r is outside the image
 * (calc_cost_sgm.cpp:152-180: x==xstart / y==ystart / x==xend-xstep), else step_1d from p-r
 * (:182-226).  S += L_r (:227-232). */
void aggregate_dir(uint32_t* S, const uint8_t* C, int W, int H, int D, int dx, int dy, int P1, int P2) {
    const int E = D + 1;
    std::vector<uint8_t> bufA((size_t)W * E), bufB((size_t)W * E);
    uint8_t* prev = bufA.data();
    uint8_t* cur = bufB.data();
    const int ys = dy >= 0 ? 1 : -1, y0 = dy >= 0 ? 0 : H - 1;
    const int xs = dx >= 0 ? 1 : -1, x0 = dx >= 0 ? 0 : W - 1;
    for (int yi = 0, y = y0; yi < H; yi++, y += ys) {
        for (int xi = 0, x = x0; xi < W; xi++, x += xs) {
            const int px = x - dx, py = y - dy;
            const bool inside = px >= 0 && px < W && py >= 0 && py < H;
            uint8_t* L = cur + (size_t)x * E;
            const uint8_t* c = C + ((size_t)y * W + x) * D;
            if (!inside) {
                memcpy(L, c, D);
                L[D] = 0;                                             /* :154,:164 -- 0, not min(C) */
            } else {
                const uint8_t* Lp = (dy == 0 ? cur : prev) + (size_t)px * E;
                step_1d(L, Lp, c, D, P1, P2);
            }
            uint32_t* s = S + ((size_t)y * W + x) * D;
            for (int d = 0; d < D; d++) s[d] += L[d];
        }
        uint8_t* t = prev; prev = cur; cur = t;
    }
}

}  // namespace

extern "C" {

/* common.cpp:3-27 */
void fsgm_oracle_census(const uint8_t* img, uint32_t* cen, int W, int H, int halfWin) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint32_t code = 0;
            const uint8_t ctr = img[x + W * y];
            for (int oy = -halfWin; oy <= halfWin; oy++)
                for (int ox = -halfWin; ox <= halfWin; ox++) {
                    const int y2 = clampi(y + oy, 0, H - 1), x2 = clampi(x + ox, 0, W - 1);
                    code = (code + (img[x2 + W * y2] >= ctr ? 1u : 0u)) << 1;   /* :19-21 */
                }
            cen[x + y * W] = code;
        }
}

/* calc_cost_sgm.cpp:319-412 */
void fsgm_oracle_epi_cost(uint8_t* C, uint8_t* Craw,
                          const uint8_t* I1, const uint8_t* I2, int W, int H, int D, double vMax,
                          const double* pixelPosD0, const double* normDir, const double* offset) {
    const size_t NP = (size_t)W * H;
    std::vector<uint32_t> cen1(NP), cen2(NP);
    fsgm_oracle_census(I1, cen1.data(), W, H, 2);          /* :328 */
    fsgm_oracle_census(I2, cen2.data(), W, H, 2);          /* :329 */
    const double* dirX = normDir;      const double* dirY = normDir + NP;        /* :333-334 */
    const double* p0X = pixelPosD0;    const double* p0Y = pixelPosD0 + NP;      /* :336-337 */
    const double n = D + 1;                                                       /* :339 */
    std::vector<uint8_t> tmp;
    uint8_t* raw = Craw;
    if (!raw) { tmp.resize(NP * D); raw = tmp.data(); }

    /* :360-361 depend on d only, so tabulate them (same fp64 expressions, same order). */
    std::vector<double> vzInd(D);
    for (int d = 0; d < D; d++) {
        const double vzRatio = 1.0 * d / n * vMax;
        vzInd[d] = vzRatio / (1 - vzRatio);
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const double bx = p0X[p] - 1, by = p0Y[p] - 1;                        /* :348-349 */
            const double ux = dirX[p], uy = dirY[p], off = offset[p];
            const uint32_t c1 = cen1[p];
            uint8_t* out = raw + p * D;
            for (int d = 0; d < D; d++) {
                const double ox = off * vzInd[d] * ux;                            /* :365 */
                const double oy = off * vzInd[d] * uy;                            /* :366 */
                int x2 = f64_to_i32(round(bx + ox));                              /* :371 */
                int y2 = f64_to_i32(round(by + oy));                              /* :372 */
                x2 = clampi(x2, 0, W - 1);
                y2 = clampi(y2, 0, H - 1);
                out[d] = (uint8_t)__builtin_popcount(c1 ^ cen2[(size_t)y2 * W + x2]);   /* :377-378 */
            }
        }
    /* :387-407  5x5 box mean, replicate border, (u8)(1.0*sum/25 + 0.5) */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int d = 0; d < D; d++) {
                unsigned sum = 0;
                for (int dy = -2; dy <= 2; dy++) {
                    const int y1 = clampi(y + dy, 0, H - 1);
                    for (int dx = -2; dx <= 2; dx++) {
                        const int x1 = clampi(x + dx, 0, W - 1);
                        sum += raw[((size_t)y1 * W + x1) * D + d];
                    }
                }
                C[((size_t)y * W + x) * D + d] = f64_to_u8(1.0 * sum / 25 + 0.5);   /* :404 */
            }
}

/* calc_cost_sgm.cpp:86-257 */
void fsgm_oracle_epi_aggregate(uint32_t* S, const uint8_t* C, int W, int H, int D,
                               int P1, int P2, int paths) {
    memset(S, 0, sizeof(uint32_t) * ((size_t)W * H * D + 1));       /* :96 (+1: see header) */
    /* pass 0 (:106-112): L1 from the left, L3 from above, L2 from above-left, L4 from above-right */
    static const int dirs[4][2] = {{1, 0}, {0, 1}, {1, 1}, {-1, 1}};
    const int nd = paths == 8 ? 4 : 2;                               /* :104 enableDiagnalPath */
    for (int pass = 0; pass < 2; pass++)                             /* :103 totalPass = 2; :115-123 mirror */
        for (int k = 0; k < nd; k++) {
            const int sgn = pass == 0 ? 1 : -1;
            aggregate_dir(S, C, W, H, D, sgn * dirs[k][0], sgn * dirs[k][1], P1, P2);
        }
}

/* calc_cost_sgm.cpp:259-308 */
void fsgm_oracle_epi_wta(uint32_t* bestD, uint32_t* minC, const uint32_t* S,
                         int W, int H, int D, int subpixel) {
    for (size_t p = 0; p < (size_t)W * H; p++) {
        const uint32_t* s = S + p * D;
        uint32_t lo = s[0], idx = 0;
        for (int d = 1; d < D; d++)
            if (s[d] < lo) { lo = s[d]; idx = d; }                   /* :267 strict: first minimum */
        minC[p] = lo;
        if (!subpixel) { bestD[p] = idx; continue; }
        if (idx > 1 && idx < (uint32_t)D) {                          /* :293 (sic: d==1 skipped, d==D-1 kept) */
            const double c_1 = (double)s[idx - 1], c = (double)s[idx], c1 = (double)s[idx + 1];
            double sub = idx;
            if (c1 < c_1) sub = sub + (c1 - c_1) / (c - c_1) / 2.0;  /* :299 */
            else          sub = sub + (c1 - c_1) / (c - c1) / 2.0;   /* :301 */
            bestD[p] = f64_to_u32(sub * 256);                        /* :303 */
        } else {
            bestD[p] = idx * 256;                                    /* :305 */
        }
    }
}

/* calc_cost_sgm.cpp:414-426 */
void fsgm_oracle_epi_vz_to_disp(uint32_t* bestD, int W, int H, const double* offset,
                                double vMax, int n) {
    for (size_t p = 0; p < (size_t)W * H; p++) {
        const double d = (double)bestD[p] / 256;
        const double vzRatio = d / n * vMax;
        const double vzInd = vzRatio / (1 - vzRatio);
        bestD[p] = f64_to_u32((offset[p] * vzInd) * 256);
    }
}

/* calc_cost_sgm.cpp:429-480 calc_disp_from_first + :482-536 forward_backward_check (USE_VZIND
 * branch).  Dead code in the reference as shipped (the call at :589-590 is commented out): it
 * would run on bestD BEFORE convert_vzInd_to_disp.  D1 = bestD (vz index * 256). */
void fsgm_oracle_epi_fb_check(uint8_t* conf, uint32_t* D2, const uint32_t* D1, int W, int H,
                              const double* pixelPosD0, const double* normDir, const double* offset,
                              double vMax, int n, int thr) {
    const size_t NP = (size_t)W * H;
    const uint32_t INVALID = 512u << 8;                              /* :5 INVALID_DISPARITY */
    const double* dirX = normDir;   const double* dirY = normDir + NP;
    const double* p0X = pixelPosD0; const double* p0Y = pixelPosD0 + NP;
    for (size_t p = 0; p < NP; p++) D2[p] = INVALID;                 /* :440-442 */
    auto displacement = [&](size_t p) {                              /* :447-451 / :502-506 */
        double d = (double)D1[p] / 256;
        const double vzRatio = d / n * vMax;
        const double vzInd = vzRatio / (1 - vzRatio);
        return offset[p] * vzInd;
    };
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const double d = displacement(p);
            const int p2x = f64_to_i32((p0X[p] - 1) + d * dirX[p]);  /* :462 truncation */
            const int p2y = f64_to_i32((p0Y[p] - 1) + d * dirY[p]);  /* :463 */
            for (int dy = 0; dy <= 1; dy++)
                for (int dx = 0; dx <= 1; dx++) {
                    const long long tx = (long long)dx + p2x, ty = (long long)dy + p2y;
                    if (tx >= 0 && tx < W && ty >= 0 && ty < H) {
                        uint32_t& t = D2[(size_t)ty * W + tx];
                        if (t == INVALID || t < D1[p]) t = D1[p];    /* :472-473 */
                    }
                }
        }
    memset(conf, 1, NP);                                             /* :485 */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const double d = displacement(p);
            const int p2x = f64_to_i32(round((p0X[p] - 1) + d * dirX[p]));   /* :516 */
            const int p2y = f64_to_i32(round((p0Y[p] - 1) + d * dirY[p]));   /* :517 */
            if (p2x < 0 || p2x > W - 1 || p2y < 0 || p2y > H - 1) { conf[p] = 0; continue; }   /* :519-522 */
            const uint32_t t = D2[(size_t)p2y * W + p2x];
            if (t == INVALID) { conf[p] = 0; continue; }                                       /* :524-527 */
            long long diff = (long long)(int32_t)D1[p] - (long long)(int32_t)t;                /* :529 int(D1) - int(D2) */
            if (diff < 0) diff = -diff;
            if (diff > thr) conf[p] = 0;
        }
}

/* calc_cost_sgm.cpp:539-598 */
void fsgm_oracle_calc_cost_sgm(uint32_t* bestD, uint32_t* minC,
                               const uint8_t* I1, const uint8_t* I2, int W, int H, int D,
                               double vMax, const double* pixelPosD0, const double* normDir,
                               const double* offset, int P1, int P2, int paths,
                               uint8_t* C_out, uint32_t* S_out) {
    const size_t N = (size_t)W * H * D;
    std::vector<uint8_t> Cbuf;
    std::vector<uint32_t> Sbuf;
    uint8_t* C = C_out;
    uint32_t* S = S_out;
    if (!C) { Cbuf.resize(N); C = Cbuf.data(); }
    if (!S) { Sbuf.resize(N + 1); S = Sbuf.data(); }
    fsgm_oracle_epi_cost(C, NULL, I1, I2, W, H, D, vMax, pixelPosD0, normDir, offset);   /* :581 */
    fsgm_oracle_epi_aggregate(S, C, W, H, D, P1, P2, paths);                              /* :584 */
    fsgm_oracle_epi_wta(bestD, minC, S, W, H, D, 1);                  /* :560 subPixelRefine = true */
    fsgm_oracle_epi_vz_to_disp(bestD, W, H, offset, vMax, D + 1);     /* :593 */
}

}  // extern "C"
