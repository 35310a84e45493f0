/*
 * fsgm_oracle_pyramid.cpp -- CPU restatement of the pyramidal driver pyramidal_sgm.m (reference:
 * /root/reference, cited as file:line) around the oracle's calc_pyd_cost_sgm.
 *
 * TEST INFRASTRUCTURE ONLY (see fsgm_oracle.h).  PARITY UNPINNED, twice over:
 *   - the level loop restates pyramidal_sgm.m line by line, but the MEX it calls is the unpinned
 *     oracle of fsgm_oracle_pyd.cpp;
 *   - impyramid, rgb2gray and imresize are MATLAB Image Processing Toolbox functions that are not
 *     part of the reference tree (and MATLAB is not in this image).  They are restated here from
 *     their published behaviour:
 *       impyramid(A,'reduce') = imresize(A, 0.5, {kernel, 5}, 'OutputSize', ceil(size/2),
 *                               'Antialiasing', false) with the piecewise-constant kernel
 *                               [.0625 .25 .375 .25 .0625]: output sample i is centred on input
 *                               sample 2i (0-based), borders are mirrored with the edge sample
 *                               repeated (imresize's index table [1:n n:-1:1]), rows are resized
 *                               first and the intermediate is rounded back to uint8;
 *       rgb2gray              = round(0.298936021293775 R + 0.587043074451121 G + 0.114020904255103 B);
 *       imresize(A,2,'nearest') = sample duplication, output sample i <- input sample floor(i/2).
 *
 * Memory order: images u8 [channel][y][x] (x fastest: what the reference's drivers hand a MEX
 * after permute([2 1 3])), flow f64 [2][y][x] with plane 0 = x.
 */
#include "fsgm_oracle.h"
#include <math.h>
#include <string.h>
#include <vector>

namespace {

inline int mirror(int i, int n) {                   /* imresize: aux = [1:n, n:-1:1]; aux(mod(i-1, 2n)+1) */
    const int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}

/* weights 1 4 6 4 1 (/16) on samples 2i-2 .. 2i+2; uint8(x) rounds half up for x >= 0 */
inline uint8_t tap5(int a, int b, int c, int d, int e) { return (uint8_t)((a + 4 * b + 6 * c + 4 * d + e + 8) >> 4); }

}  // namespace

extern "C" {

void fsgm_oracle_impyramid_reduce(uint8_t* out, const uint8_t* in, int W, int H) {
    const int W2 = (W + 1) / 2, H2 = (H + 1) / 2;   /* outputSize = ceil([M N]/2) */
    std::vector<uint8_t> tmp((size_t)H2 * W);
    for (int i = 0; i < H2; i++)                    /* MATLAB dimension 1 (rows) first */
        for (int x = 0; x < W; x++) {
            const uint8_t* c = in + x;
            tmp[(size_t)i * W + x] = tap5(c[(size_t)mirror(2 * i - 2, H) * W], c[(size_t)mirror(2 * i - 1, H) * W],
                                          c[(size_t)mirror(2 * i, H) * W], c[(size_t)mirror(2 * i + 1, H) * W],
                                          c[(size_t)mirror(2 * i + 2, H) * W]);
        }
    for (int i = 0; i < H2; i++)
        for (int j = 0; j < W2; j++) {
            const uint8_t* r = &tmp[(size_t)i * W];
            out[(size_t)i * W2 + j] = tap5(r[mirror(2 * j - 2, W)], r[mirror(2 * j - 1, W)], r[mirror(2 * j, W)],
                                           r[mirror(2 * j + 1, W)], r[mirror(2 * j + 2, W)]);
        }
}

void fsgm_oracle_rgb2gray(uint8_t* out, const uint8_t* rgb, int W, int H) {
    const size_t n = (size_t)W * H;
    for (size_t i = 0; i < n; i++) {
        const double v = 0.298936021293775 * rgb[i] + 0.587043074451121 * rgb[n + i] + 0.114020904255103 * rgb[2 * n + i];
        out[i] = (uint8_t)floor(v + 0.5);           /* < 255.5 always */
    }
}

/* pyramidal_sgm.m:1-77.  I0, I1: u8 [channels][H][W], channels 1 or 3.  mv: f64 [2][H][W] (level 1,
 * mvCurLevel); minC u32 [H][W] (level 1); mvPyd (may be NULL): numPyd pointers, entry l-1 receives
 * level l's flow f64 [2][H_l][W_l] when not NULL. */
void fsgm_oracle_pyramidal_sgm(double* mv, uint32_t* minC, double** mvPyd,
                               const uint8_t* I0, const uint8_t* I1, int W, int H, int channels, int numPyd,
                               int P1, int P2, int aggHalfWinSize, int verSearchHalfWinSize, int horSearchHalfWinSize,
                               int enableDiagonal, int totalPass, int adaptiveP2) {
    std::vector<int> Ws(numPyd), Hs(numPyd);
    std::vector<std::vector<uint8_t>> p0(numPyd), p1(numPyd);
    Ws[0] = W; Hs[0] = H;
    p0[0].assign(I0, I0 + (size_t)channels * W * H);
    p1[0].assign(I1, I1 + (size_t)channels * W * H);
    for (int l = 1; l < numPyd; l++) {              /* :28-31 impyramid 'reduce', channel by channel */
        Ws[l] = (Ws[l - 1] + 1) / 2; Hs[l] = (Hs[l - 1] + 1) / 2;
        const size_t n = (size_t)Ws[l] * Hs[l], np = (size_t)Ws[l - 1] * Hs[l - 1];
        p0[l].resize(channels * n); p1[l].resize(channels * n);
        for (int c = 0; c < channels; c++) {
            fsgm_oracle_impyramid_reduce(&p0[l][c * n], &p0[l - 1][c * np], Ws[l - 1], Hs[l - 1]);
            fsgm_oracle_impyramid_reduce(&p1[l][c * n], &p1[l - 1][c * np], Ws[l - 1], Hs[l - 1]);
        }
    }
    const int Sy = 2 * verSearchHalfWinSize + 1;
    int mvW = Ws[numPyd - 1], mvH = Hs[numPyd - 1];
    std::vector<double> mvPre((size_t)2 * mvW * mvH, 0.0);                       /* :34 */
    for (int l = numPyd - 1; l >= 0; l--) {                                      /* :37 */
        const int w = Ws[l], h = Hs[l];
        const size_t n = (size_t)w * h;
        std::vector<uint8_t> g0(n), g1(n);
        if (channels == 3) {                                                     /* :44-45 */
            fsgm_oracle_rgb2gray(g0.data(), p0[l].data(), w, h);
            fsgm_oracle_rgb2gray(g1.data(), p1[l].data(), w, h);
        } else {
            g0 = p0[l]; g1 = p1[l];
        }
        std::vector<uint32_t> bestD(n), mc(n);
        std::vector<double> mvSub(2 * n);
        fsgm_oracle_calc_pyd_cost_sgm(bestD.data(), mc.data(), mvSub.data(), g0.data(), g1.data(), w, h,       /* :50 */
                                      mvPre.data(), mvW, mvH, horSearchHalfWinSize, verSearchHalfWinSize, aggHalfWinSize,
                                      l == 0, P1, P2, enableDiagonal, totalPass, adaptiveP2, nullptr, nullptr);
        std::vector<double> cur(2 * n);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const size_t i = (size_t)y * w + x, ip = (size_t)y * mvW + x;
                const int sx = (int)(bestD[i] / Sy), sy = (int)(bestD[i] % Sy);                                 /* :57 ind2sub */
                const double mx = (double)(sx - horSearchHalfWinSize), my = (double)(sy - verSearchHalfWinSize);   /* :59-60 */
                cur[i] = (mx + mvPre[ip]) + mvSub[i];                                                           /* :64 */
                cur[n + i] = (my + mvPre[(size_t)mvW * mvH + ip]) + mvSub[n + i];
            }
        if (mvPyd && mvPyd[l]) memcpy(mvPyd[l], cur.data(), 2 * n * sizeof(double));                            /* :66 */
        if (l == 0) {
            memcpy(mv, cur.data(), 2 * n * sizeof(double));
            memcpy(minC, mc.data(), n * sizeof(uint32_t));
        } else {                                                                                                /* :72 2*imresize(mv,2,'nearest') */
            mvW = 2 * w; mvH = 2 * h;
            mvPre.assign((size_t)2 * mvW * mvH, 0.0);
            for (int c = 0; c < 2; c++)
                for (int y = 0; y < mvH; y++)
                    for (int x = 0; x < mvW; x++)
                        mvPre[((size_t)c * mvH + y) * mvW + x] = 2.0 * cur[c * n + (size_t)(y / 2) * w + x / 2];
        }
    }
}

}  // extern "C"
