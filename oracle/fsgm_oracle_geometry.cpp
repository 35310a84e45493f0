/*
 * fsgm_oracle_geometry.cpp -- CPU restatement of the dense half of the epipolar driver:
 * rotation_motion.m, epipolar_geometry.m:99-115 and the tail of epipolar_sgm_of.m (:33-51), around the
 * oracle's calc_cost_sgm (reference: /root/reference, cited as file:line).
 *
 * TEST INFRASTRUCTURE ONLY (see fsgm_oracle.h).  PARITY UNPINNED: MATLAB scripts, no MATLAB here; the
 * 3x3 matrix-vector products are written left to right (MATLAB hands them to BLAS, whose summation
 * order is unspecified); rgb2gray as in fsgm_oracle_pyramid.cpp.  The sparse half of
 * epipolar_geometry.m (:30-96, toolbox feature matching and LMedS) is not restated: its results
 * F, H, epipole, direction are inputs.
 */
#include "fsgm_oracle.h"
#include <math.h>
#include <vector>

extern "C" {

/* F, Hm row-major 3x3.  Pd0, nd, rflow: [2][H][W]; off: [H][W]. */
void fsgm_oracle_epipolar_maps(double* Pd0, double* nd, double* off, double* rflow, const double* F, const double* Hm,
                               double ex, double ey, int direction, int W, int H) {
    const size_t NP = (size_t)W * H;
    for (int yi = 0; yi < H; yi++)
        for (int xi = 0; xi < W; xi++) {
            const size_t p = (size_t)yi * W + xi;
            const double x = xi, y = yi;                                          /* rotation_motion.m:11-13 */
            double l[3], q[3];
            for (int i = 0; i < 3; i++) {
                l[i] = (F[3 * i] * x + F[3 * i + 1] * y) + F[3 * i + 2];          /* :49 l2 = F*p1 */
                q[i] = (Hm[3 * i] * x + Hm[3 * i + 1] * y) + Hm[3 * i + 2];       /* :21 P1 = H*P0 */
            }
            double nf = sqrt(l[0] * l[0] + l[1] * l[1]);                          /* :50 */
            if (nf < 1e-6) nf = 1.0;                                              /* :51 */
            for (int i = 0; i < 3; i++) l[i] = l[i] / nf;                         /* :52 */
            const double p1x = q[0] / q[2], p1y = q[1] / q[2], p1z = q[2] / q[2]; /* :22 */
            double ox = p1x - x, oy = p1y - y;                                    /* :23 */
            const double coef = -((l[0] * p1x + l[1] * p1y) + l[2] * p1z);        /* :27 */
            ox = ox + coef * l[0];                                                /* :28 */
            oy = oy + coef * l[1];
            rflow[p] = ox; rflow[NP + p] = oy;
            const double pdx = (x + 1.0) + ox, pdy = (y + 1.0) + oy;              /* epipolar_geometry.m:99-106 */
            Pd0[p] = pdx; Pd0[NP + p] = pdy;
            double dx = pdx - ex, dy = pdy - ey;                                  /* :107 */
            if (direction) { dx = -dx; dy = -dy; }                                /* :108-110 */
            const double len = sqrt(dx * dx + dy * dy);                           /* :112 */
            off[p] = len;
            nd[p] = dx / len; nd[NP + p] = dy / len;                              /* :113,:116-118 */
        }
}

/* epipolar_sgm_of.m:33-51 with the geometry given.  I0/I1 u8 [channels][H][W]; flow f64 [3][H][W]. */
void fsgm_oracle_epipolar_sgm_of(double* flow, uint32_t* minC, const uint8_t* I0, const uint8_t* I1, int W, int H,
                                 int channels, const double* F, const double* Hm, double ex, double ey, int direction,
                                 int dMax, double vMax, int paths) {
    const size_t NP = (size_t)W * H;
    std::vector<double> Pd0(2 * NP), nd(2 * NP), off(NP), rflow(2 * NP);
    fsgm_oracle_epipolar_maps(Pd0.data(), nd.data(), off.data(), rflow.data(), F, Hm, ex, ey, direction, W, H);   /* :24 */
    std::vector<uint8_t> g0(NP), g1(NP);
    if (channels == 3) { fsgm_oracle_rgb2gray(g0.data(), I0, W, H); fsgm_oracle_rgb2gray(g1.data(), I1, W, H); }   /* :35-38 */
    else { g0.assign(I0, I0 + NP); g1.assign(I1, I1 + NP); }
    std::vector<uint32_t> bestD(NP), mc(NP);
    fsgm_oracle_calc_cost_sgm(bestD.data(), mc.data(), g0.data(), g1.data(), W, H, dMax, vMax, Pd0.data(), nd.data(),
                              off.data(), 6, 64, paths, nullptr, nullptr);                                          /* :19,:45 */
    for (size_t p = 0; p < NP; p++) {
        const double disp = (double)bestD[p] / 256.0;                             /* :46 */
        flow[p] = disp * nd[p] + rflow[p];                                        /* :49-50 */
        flow[NP + p] = disp * nd[NP + p] + rflow[NP + p];
        flow[2 * NP + p] = 1.0;                                                   /* :51 */
        if (minC) minC[p] = mc[p];
    }
}

}  // extern "C"
