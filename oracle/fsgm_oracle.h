/*
 * fsgm_oracle.h -- CPU restatement of fSGM's cost-volume + SGM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fsgm_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * PARITY STATUS
 *   census()                       : PINNED against the reference's own common.cpp,
 *                                    compiled unmodified into oracle/_ref/ (see Makefile).
 *   everything else in this header : PARITY UNPINNED.  The four MEX sources need MATLAB's
 *                                    mex.h, which this image does not have, so they cannot be
 *                                    built here without a stand-in header, and the reference
 *                                    ships no golden vectors.  These functions restate the
 *                                    reference line by line (citations at each function).
 *
 * Memory order is the reference's native order: images [y][x] (x fastest),
 * volumes [y][x][d] (d fastest), two-plane maps plane-major (plane 0 = x).
 * All double->integer conversions follow x86-64 gcc semantics (cvttsd2si),
 * implemented explicitly so this file has no undefined behaviour.
 */
#ifndef FSGM_ORACLE_H
#define FSGM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* common.cpp:3-27 */
void fsgm_oracle_census(const uint8_t* img, uint32_t* cen, int W, int H, int halfWin);

/* calc_cost_sgm.cpp:319-412 (USE_VZIND branch).  Craw (optional, may be NULL) receives the
 * un-box-filtered Hamming costs (the reference's Ctmp). */
void fsgm_oracle_epi_cost(uint8_t* C, uint8_t* Craw,
                          const uint8_t* I1, const uint8_t* I2, int W, int H, int D, double vMax,
                          const double* pixelPosD0, const double* normDir, const double* offset);

/* calc_cost_sgm.cpp:33-66,86-257: multi-path aggregation.  paths = 4 (as shipped,
 * enableDiagnalPath=false) or 8.  S (u32 [H*W*D + 1]) receives the summed path costs; the one
 * extra trailing element is written 0 (see sgm_wta below). */
void fsgm_oracle_epi_aggregate(uint32_t* S, const uint8_t* C, int W, int H, int D,
                               int P1, int P2, int paths);

/* calc_cost_sgm.cpp:259-308: WTA + fixed-point parabola.  S must have one readable element past
 * the end (the reference reads it for the last pixel when best==D-1; we define it as 0). */
void fsgm_oracle_epi_wta(uint32_t* bestD, uint32_t* minC, const uint32_t* S,
                         int W, int H, int D, int subpixel);

/* calc_cost_sgm.cpp:414-426 */
void fsgm_oracle_epi_vz_to_disp(uint32_t* bestD, int W, int H, const double* offset,
                                double vMax, int n);

/* calc_cost_sgm.cpp:429-536: forward-backward consistency check (dead code in the shipped
 * reference: its call at :589-590 is commented out).  Runs on bestD before vz->disparity.
 * conf u8 [H*W] (1 = consistent), D2 u32 [H*W]. */
void fsgm_oracle_epi_fb_check(uint8_t* conf, uint32_t* D2, const uint32_t* D1, int W, int H,
                              const double* pixelPosD0, const double* normDir, const double* offset,
                              double vMax, int n, int thr);

/* calc_cost_sgm.cpp:539-598: the whole MEX.  paths: 4 = as shipped.  Optional debug outputs C
 * (u8 [H*W*D]) and S (u32 [H*W*D+1]) may be NULL. */
void fsgm_oracle_calc_cost_sgm(uint32_t* bestD, uint32_t* minC,
                               const uint8_t* I1, const uint8_t* I2, int W, int H, int D,
                               double vMax, const double* pixelPosD0, const double* normDir,
                               const double* offset, int P1, int P2, int paths,
                               uint8_t* C_out, uint32_t* S_out);

/* ---- pyramidal 2-D variant: calc_pyd_cost_sgm.cpp ---- */

/* calc_pyd_cost_sgm.cpp:374-437 */
void fsgm_oracle_pyd_cost(uint8_t* C, const uint32_t* cen1, const uint32_t* cen2, int W, int H,
                          const double* preMv, int mvW, int mvH, int rAgg, int rX, int rY);

/* calc_pyd_cost_sgm.cpp:34-89,114-296.  S: u32 [H*W*Sx*Sy]. */
void fsgm_oracle_pyd_aggregate(uint32_t* S, const uint8_t* I1, const uint8_t* C, int W, int H,
                               const double* preMv, int mvW, int mvH, int Sx, int Sy,
                               int P1, int P2, int diagonal, int totalPass, int adaptiveP2);

/* calc_pyd_cost_sgm.cpp:298-364 */
void fsgm_oracle_pyd_wta(uint32_t* bestD, uint32_t* minC, double* mvSub, const uint32_t* S,
                         int W, int H, int Sx, int Sy, int subpixel);

/* calc_pyd_cost_sgm.cpp:439-510: whole MEX. */
void fsgm_oracle_calc_pyd_cost_sgm(uint32_t* bestD, uint32_t* minC, double* mvSub,
                                   const uint8_t* I1, const uint8_t* I2, int W, int H,
                                   const double* preMv, int mvW, int mvH,
                                   int rX, int rY, int rAgg, int subpixel, int P1, int P2,
                                   int diagonal, int totalPass, int adaptiveP2,
                                   uint8_t* C_out, uint32_t* S_out);

/* ---- pyramidal driver: pyramidal_sgm.m (fsgm_oracle_pyramid.cpp; the MATLAB toolbox functions it
 * calls -- impyramid, rgb2gray, imresize -- are restated from their published behaviour) ---- */
void fsgm_oracle_impyramid_reduce(uint8_t* out, const uint8_t* in, int W, int H);   /* out: ceil(W/2) x ceil(H/2) */
void fsgm_oracle_rgb2gray(uint8_t* out, const uint8_t* rgb, int W, int H);          /* rgb: [3][H][W] */
/* pyramidal_sgm.m:1-77.  I0/I1 u8 [channels][H][W]; mv f64 [2][H][W]; minC u32 [H][W]; mvPyd NULL or
 * numPyd pointers (entry l-1 = level l flow, f64 [2][H_l][W_l], H_l = ceil(H_{l-1}/2)). */
void fsgm_oracle_pyramidal_sgm(double* mv, uint32_t* minC, double** mvPyd,
                               const uint8_t* I0, const uint8_t* I1, int W, int H, int channels, int numPyd,
                               int P1, int P2, int aggHalfWinSize, int verSearchHalfWinSize, int horSearchHalfWinSize,
                               int enableDiagonal, int totalPass, int adaptiveP2);

/* ---- dense half of the epipolar driver (fsgm_oracle_geometry.cpp): rotation_motion.m,
 * epipolar_geometry.m:99-115, epipolar_sgm_of.m:33-51.  F, Hm row-major 3x3. ---- */
void fsgm_oracle_epipolar_maps(double* Pd0, double* nd, double* off, double* rflow, const double* F, const double* Hm,
                               double ex, double ey, int direction, int W, int H);
void fsgm_oracle_epipolar_sgm_of(double* flow, uint32_t* minC, const uint8_t* I0, const uint8_t* I1, int W, int H,
                                 int channels, const double* F, const double* Hm, double ex, double ey, int direction,
                                 int dMax, double vMax, int paths);

/* ---- post-processing chain of test.m:45-50 (fsgm_oracle_post.cpp): maps f64 [H][W], NaN = invalid ---- */
void fsgm_oracle_vzind2disp(double* D, const double* w, const double* O, int n_px, double vMax, double n);   /* vzInd2Disp.m */
void fsgm_oracle_speckle_filter(double* out, int32_t* labels_out, const double* image, int W, int H,
                                double maxDiff, double maxSpeckleSize);                                     /* speckle_filter.m */
void fsgm_oracle_calc_disp_from_first(double* D2, const double* D1, int W, int H, const double* Pd0,
                                      const double* nd, const double* O, double vMax, double n);            /* calc_disp_from_first.m */
void fsgm_oracle_forward_backward_check(double* out, const double* D1, const double* D2, int W, int H,
                                        const double* Pd0, const double* nd, const double* O, double vMax, double n);
void fsgm_oracle_scanline_in_fill(double* out, const double* in, int W, int H);                             /* scanline_in_fill.m */
void fsgm_oracle_vmf(double* out, const double* flow, int W, int H, int channels);                           /* vmf.m (medfilt2 5x5) */
void fsgm_oracle_postprocess(double* filterD1, double* filterD2, double* disp, const double* D1, int W, int H,
                             const double* Pd0, const double* nd, const double* O, double vMax, double n, double dMax);

/* ---- neighbour-guided candidate-list variant: calc_pyd_cost_sgm_ng.cpp ---- */
typedef struct { int32_t mvx, mvy, cost; } fsgm_oracle_cand;   /* calc_pyd_cost_sgm_ng.cpp:32-37 */

/* calc_pyd_cost_sgm_ng.cpp:370-446 */
void fsgm_oracle_ng_cost(fsgm_oracle_cand* C, const uint32_t* cen1, const uint32_t* cen2,
                         int W, int H, const double* preMv, int mvW, int mvH,
                         int rAgg, int rX, int rY);
/* calc_pyd_cost_sgm_ng.cpp:39-78,101-299 */
void fsgm_oracle_ng_aggregate_wta(uint32_t* minC, double* flow, uint32_t* S_out,
                                  const fsgm_oracle_cand* C, int W, int H, int D, int P1, int P2);
/* calc_pyd_cost_sgm_ng.cpp:308-368 */
void fsgm_oracle_ng_subpixel(double* flow, const uint32_t* cen1, const uint32_t* cen2, int W, int H);
/* calc_pyd_cost_sgm_ng.cpp:448-523: whole MEX. */
void fsgm_oracle_calc_pyd_cost_sgm_ng(uint32_t* minC, double* flow,
                                      const uint8_t* I1, const uint8_t* I2, int W, int H,
                                      const double* preMv, int mvW, int mvH,
                                      double halfSearchWinSize, double aggSize, int subpixel,
                                      int P1, int P2, fsgm_oracle_cand* C_out, uint32_t* S_out);

/* ---- on-the-fly neighbour-guided variant: calc_cost_sgm_ng.cpp ---- */
/* calc_cost_sgm_ng.cpp:46-98,122-186,188-419,484-526.  rand_stream: the sequence the reference
 * would draw from libc rand() (2 draws per random hint, raster order); n_rand entries. */
void fsgm_oracle_calc_cost_sgm_ng(uint32_t* minC, double* flow,
                                  const uint8_t* I1, const uint8_t* I2, int W, int H,
                                  int P1, int P2, const int32_t* rand_stream, int64_t n_rand);
int64_t fsgm_oracle_sgm_ng_rand_draws(int W, int H);

#ifdef __cplusplus
}
#endif
#endif
