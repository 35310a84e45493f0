/*
 * fsgm.h -- C ABI of libfsgm_hip.so: fSGM's matching-cost + multi-path SGM hot path on
 * AMD Instinct MI355X (gfx950, hand-written HIP).
 *
 * This is the drop-in boundary.  Every entry point below replaces one function of the
 * reference's MEX layer (file:line cited per function, relative to the fSGM tree).  The four
 * mexFunction gateways in fsgm_amd/mex/ are thin shims over the *_host entry points: host
 * pointers in, host pointers out, plain C types only, nothing of torch/HIP in a signature
 * (streams travel as void*).
 *
 * Memory order is the reference's native order, i.e. what MATLAB hands a MEX after the
 * drivers' permute([2 1 3]) (epipolar_sgm_of.m:33-43, pyramidal_sgm.m:44-46, ng_sgm.m:15-17):
 *   images  u8  [height][width]        x fastest      (MATLAB: width x height, column-major)
 *   maps    f64 [plane][height][width] plane 0 = x    (MATLAB: width x height x 2)
 *   volumes u8  [height][width][D]     d fastest      (calc_cost_sgm.cpp:345,389)
 *
 * There is NO CPU fallback in this library: every compute entry point needs a HIP device and
 * fails with FSGM_ERR_HIP when there is none.
 */
#ifndef FSGM_H
#define FSGM_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FSGM_VERSION_MAJOR 0
#define FSGM_VERSION_MINOR 1

typedef enum {
    FSGM_OK = 0,
    FSGM_ERR_INVALID = 1,      /* bad argument (null pointer, size, class) -- the reference validates nothing */
    FSGM_ERR_HIP = 2,          /* HIP runtime error / no device */
    FSGM_ERR_NOMEM = 3,
    FSGM_ERR_UNSUPPORTED = 4
} fsgm_status;

/* Thread-local text of the last error returned on this thread. */
const char* fsgm_last_error(void);
/* Number of HIP devices visible (0 if none / no runtime). */
int fsgm_device_count(void);
/* "gfx950"-style arch string of a device, written into buf. */
fsgm_status fsgm_device_arch(int device, char* buf, size_t buflen);
/* Free every cached plan / HBM buffer held for the *_host entry points (mexAtExit hook). */
void fsgm_shutdown(void);

/* ------------------------------------------------------------------------------------------
 * calc_cost_sgm  (calc_cost_sgm.cpp:539-598, called from epipolar_sgm_of.m:45)
 * ------------------------------------------------------------------------------------------ */

/* Run-time form of the reference's compile-time switches.  fsgm_epi_params_default() returns
 * the shipped values. */
typedef struct {
    int32_t paths;        /* 4 = as shipped (enableDiagnalPath=false, calc_cost_sgm.cpp:104); 8 = with diagonals */
    int32_t subpixel;     /* 1 = as shipped (subPixelRefine=true, calc_cost_sgm.cpp:560) */
    int32_t vz_to_disp;   /* 1 = as shipped (USE_VZIND, calc_cost_sgm.cpp:4,592-594) */
    int32_t device;       /* HIP device ordinal */
    int32_t fb_check;     /* 0 = as shipped (call commented out, calc_cost_sgm.cpp:589-590: conf and bestD2 stay 0);
                             1 = run forward_backward_check (:482-536, threshold 2) on bestD before vz->disparity.
                             Needs dMax <= 511 (bestD must stay below INVALID_DISPARITY = 512<<8) */
} fsgm_epi_params;

fsgm_epi_params fsgm_epi_params_default(void);

typedef struct {
    const uint8_t* I1;            /* prhs[0]  u8 [H][W] */
    const uint8_t* I2;            /* prhs[1] */
    int32_t width, height;        /* mxGetM / mxGetN of prhs[0] (calc_cost_sgm.cpp:562-563) */
    int32_t dMax;                 /* prhs[2] */
    double  vMax;                 /* prhs[3] */
    const double* pixelPosD0;     /* prhs[4]  f64 [2][H][W], 1-based coordinates */
    const double* normDir;        /* prhs[5]  f64 [2][H][W] */
    const double* offset;         /* prhs[6]  f64 [H][W] */
    int32_t P1, P2;               /* prhs[7], prhs[8] (truncated to int like mxGetScalar -> int) */
} fsgm_epi_in;

typedef struct {
    uint32_t* bestD;              /* plhs[0]  u32 [H][W], disparity * 256 */
    uint32_t* minC;               /* plhs[1]  u32 [H][W] */
    /* optional debug taps (NULL = not wanted): the cost volume C (u8 [H][W][D]) and the summed
     * path costs S (u32 [H][W][D]) -- internal arrays of the reference (calc_cost_sgm.cpp:579,95) */
    uint8_t*  C;
    uint32_t* S;
    /* plhs[2], plhs[3]: filled only when fb_check = 1 and the pointer is not NULL (the caller's
     * zero-initialised arrays stay untouched otherwise, like in the shipped reference) */
    uint8_t*  conf;               /* u8  [H][W], 1 = forward-backward consistent */
    uint32_t* bestD2;             /* u32 [H][W], disparity of the second view, 512<<8 where invalid */
} fsgm_epi_out;

/* One frame, host pointers in / host pointers out.  This is what the calc_cost_sgm gateway
 * calls. */
fsgm_status fsgm_calc_cost_sgm_host(const fsgm_epi_in* in, const fsgm_epi_out* out,
                                    const fsgm_epi_params* prm);

/* A batch of independent frames of identical shape, processed concurrently on one device. */
fsgm_status fsgm_calc_cost_sgm_batch_host(int32_t n_frames, const fsgm_epi_in* in,
                                          const fsgm_epi_out* out, const fsgm_epi_params* prm);

/* ---- device-resident plan: buffers for `batch` frames stay in HBM across calls ---- */
typedef struct fsgm_epi_plan fsgm_epi_plan;

/* stage bits for fsgm_epi_plan_run / _time */
#define FSGM_STAGE_COST      1   /* census x2 + Hamming cost fill + 5x5 box   (calc_cost_sgm.cpp:319-412) */
#define FSGM_STAGE_AGGREGATE 2   /* multi-path DP: C -> per-path L_r          (calc_cost_sgm.cpp:86-257)  */
#define FSGM_STAGE_WTA       4   /* sum of paths, argmin, parabola, vz->disp  (calc_cost_sgm.cpp:259-308,414-426) */
#define FSGM_STAGE_ALL       7

fsgm_status fsgm_epi_plan_create(fsgm_epi_plan** plan, int32_t width, int32_t height,
                                 int32_t dMax, int32_t batch, const fsgm_epi_params* prm);
void        fsgm_epi_plan_destroy(fsgm_epi_plan* plan);
fsgm_status fsgm_epi_plan_set_penalties(fsgm_epi_plan* plan, int32_t P1, int32_t P2, double vMax);
/* Aggregation strategy: 0 = auto, 1 = the per-direction line kernels, 2 = the fused pipeline whenever eligible
 * (8 paths: horizontal pair + down sweep + final up sweep with the WTA inside -- D = 16<<k, no-wrap penalties with
 * P1 <= P2 and 3*(P1+P2) <= 255; the shipped 4 paths: the pair kernels -- 2*(P1+P2) <= 255), 3 = 8 paths only: the down and the
 * up sweep side by side and a WTA kernel over the three sums (half the latency of 2, 3 B per voxel more traffic),
 * 4 = the band sweeps (epi_band.hip: all four paths of a raster pass in one sweep, one workgroup per frame, for batches of
 * hundreds of frames; D = 16<<k, no-wrap penalties with P1 <= P2, P1 + P2 <= 127), 5 = the band sweeps with the bands of a
 * frame as workgroups of their own that hand their last row over while they run ("band16chain/nowrap"),
 * 6 = 8 paths only: as 3, but the two sweeps meet in the middle -- each writes its sum for its first half of the rows and crosses
 * the other's half as a final sweep with the WTA inside ("sweep16mid/nowrap": the traffic of 2 on the chain of 3).
 * Auto picks by batch size, frame shape and path count; fsgm_epi_auto_pipeline() below answers for any configuration and
 * fsgm_epi_plan_kernel_name() for a plan.  At 1242x375x128, 256 CUs: 8 paths -- line kernels below 4 frames, 3 below 10, 6 below 26,
 * 2 up to ~229, then 4 where a round of one workgroup per frame pays (230-256, 473-512, ...) and 5 between those rounds; 4 paths --
 * line kernels below 9 frames, then 2, then 4 / 5 likewise.  The switch points move with the frame shape by voxels^(-2/3)
 * (FSGM_EPI_SHAPE_SCALE).  Results are identical in every mode.
 * HBM a plan holds per frame beyond C (allocated when a mode first runs, kept until the plan is destroyed; N = W*H*D bytes):
 * mode 1: paths x N (path volumes); mode 2: 2 N + N/8 + boundary states (8 paths) / N + 2 N/8 (4 paths); modes 3, 6: one more N;
 * modes 4, 5: N + N/4 (9th-bit plane, 8 paths only) + one hand-off map of 3*W*D bytes (mode 5: one per band boundary,
 * ~1.4 B per voxel at 1242x375x128), + 10 bytes per pixel of WTA records for modes 2-5.  Mode 5 synchronises its
 * workgroups through bounded polls on device memory; a poll that gives up raises a flag that fsgm_epi_plan_sync / _download /
 * _time report as FSGM_ERR_HIP.  The cost stage's own buffers (images, census, coordinate maps) appear with the first upload
 * or FSGM_STAGE_COST run: an aggregation-only plan never allocates them. */
fsgm_status fsgm_epi_plan_set_agg_mode(fsgm_epi_plan* plan, int32_t mode);
/* host -> HBM (async on the plan's stream) */
fsgm_status fsgm_epi_plan_upload(fsgm_epi_plan* plan, int32_t frame, const uint8_t* I1,
                                 const uint8_t* I2, const double* pixelPosD0,
                                 const double* normDir, const double* offset);
/* aggregation-only use: put a ready cost volume into slot `frame` (skips FSGM_STAGE_COST) */
fsgm_status fsgm_epi_plan_upload_cost(fsgm_epi_plan* plan, int32_t frame, const uint8_t* C);
/* resident cost volume of frame dst <- frame src with the columns rotated by roll_cols (dst[y][(x + roll) % W] = src[y][x]),
 * device to device on the plan's stream: fills a large batch with distinct volumes without a PCIe transfer each (bench.py) */
fsgm_status fsgm_epi_plan_copy_cost(fsgm_epi_plan* plan, int32_t dst, int32_t src, int32_t roll_cols);
/* offset map only (needed by the vz->disp step when the cost stage is skipped) */
fsgm_status fsgm_epi_plan_upload_offset(fsgm_epi_plan* plan, int32_t frame, const double* offset);
fsgm_status fsgm_epi_plan_run(fsgm_epi_plan* plan, int32_t stages);
fsgm_status fsgm_epi_plan_sync(fsgm_epi_plan* plan);
/* HBM -> host (synchronous) */
fsgm_status fsgm_epi_plan_download(fsgm_epi_plan* plan, int32_t frame, uint32_t* bestD, uint32_t* minC);
fsgm_status fsgm_epi_plan_download_fb(fsgm_epi_plan* plan, int32_t frame, uint8_t* conf, uint32_t* bestD2);
fsgm_status fsgm_epi_plan_download_cost(fsgm_epi_plan* plan, int32_t frame, uint8_t* C);
fsgm_status fsgm_epi_plan_download_sum(fsgm_epi_plan* plan, int32_t frame, uint32_t* S);
/* debug tap: the census codes of the two images as FSGM_STAGE_COST left them (common.cpp:3-27; u32 [H][W],
 * the reference's bit order: first tap at bit 25, bit 0 always 0); either pointer may be NULL */
fsgm_status fsgm_epi_plan_download_census(fsgm_epi_plan* plan, int32_t frame, uint32_t* cen1, uint32_t* cen2);
/* Average milliseconds of one fsgm_epi_plan_run(stages) over `iters` back-to-back runs after
 * `warmup` untimed ones, measured with HIP events on the stream the kernels run on. */
fsgm_status fsgm_epi_plan_time(fsgm_epi_plan* plan, int32_t stages, int32_t warmup,
                               int32_t iters, float* ms_avg);
/* the hipStream_t the plan launches on */
void*       fsgm_epi_plan_stream(fsgm_epi_plan* plan);
/* which aggregation kernel the plan selected: "band16/nowrap" (band sweeps, one workgroup per frame: hundreds of frames),
 * "band16chain/nowrap" (band sweeps, the bands of a frame as workgroups of their own), "sweep16/nowrap" (8 paths, block sweep
 * pipeline), "sweep16par/nowrap" (8 paths, parallel sweeps: auto mode for 4..17 frames), "pairs16/nowrap" (4 paths, pair
 * pipeline), "packed16/nowrap", "packed16/wrap" (per-direction line kernels), "generic" (any dMax).  New names may be added:
 * dispatch on these with a default branch. */
const char* fsgm_epi_plan_kernel_name(fsgm_epi_plan* plan);
/* The same answer without a plan: what auto mode takes for `batch` frames of width x height x dMax with these penalties, costs
 * up to cmax (24 for volumes built by the cost stage) and a device of `cus` compute units (MI355X: 256).  The switch points
 * were measured at 1242x375x128 and move with the frame's voxels^(-2/3) (smaller frames stay on the line kernels for longer);
 * FSGM_EPI_PAR_MIN / _PAR_MAX / _PAIRS_MIN / _BAND_MIN in the environment override them.  "" for invalid arguments. */
const char* fsgm_epi_auto_pipeline(int32_t width, int32_t height, int32_t dMax, int32_t batch, int32_t paths, int32_t P1, int32_t P2,
                                   int32_t cmax, int32_t cus);
/* device-to-device copy bandwidth probe (GB/s, read+written bytes counted) used by bench.py: the library's own
 * grid-stride copy kernel, 16 B per lane per access -- the access width of the aggregation kernels */
fsgm_status fsgm_measure_copy_bandwidth(int32_t device, size_t bytes, int32_t iters, double* gbps);
/* mode 0: the same; mode 1: hipMemcpyAsync device-to-device (the runtime's blit kernel), for comparison */
fsgm_status fsgm_measure_copy_bandwidth2(int32_t device, size_t bytes, int32_t iters, int32_t mode, double* gbps);

/* sgm(C, P1, P2) -- the call shape of sgm.m:1 / test.m:36 ("perform SGM on cost volume C"), served by the calc_cost_sgm
 * path's aggregation + WTA (calc_cost_sgm.cpp:86-316): C u8 [H][W][dMax] (d fastest) in; bestD u32 [H][W] = disparity
 * index * 256 with the MEX's parabola (no vz conversion), minC u32 [H][W], optionally S u32 [H][W][dMax] out.
 * SEMANTICS ARE THE MEX'S, NOT sgm.m's: path arithmetic is mod 256 where sgm.m saturates (MATLAB uint8), a path's
 * first pixel stores minimum 0 (calc_cost_sgm.cpp:154) where sgm.m takes min(C), and d = 1 is never refined
 * (SURVEY 8(a) note on sgm.m) -- results differ from sgm.m's from the second pixel of every path on.
 * paths: 4 (sgm.m's enableDiagonalPath = false) or 8. */
fsgm_status fsgm_sgm_host(const uint8_t* C, int32_t width, int32_t height, int32_t dMax, int32_t P1, int32_t P2,
                          int32_t paths, uint32_t* bestD, uint32_t* minC, uint32_t* S, int32_t device);

/* census(img, cen, width, height) of common.cpp:3-27 on its own: u8 [H][W] -> u32 [H][W], 5x5 window, replicate
 * border, `neighbour >= centre`, first tap at bit 25, trailing shift (bit 0 = 0).  Any width/height >= 1. */
fsgm_status fsgm_census_host(const uint8_t* img, int32_t width, int32_t height, uint32_t* cen, int32_t device);

/* ------------------------------------------------------------------------------------------
 * epipolar_sgm_of with the dense maps made on the device  (dense half only)
 *
 * epipolar_geometry.m has a sparse half -- SURF features, an LMedS fundamental matrix, two SVDs and
 * the expansion vote (:30-96) -- the toolbox part of it (:130-149) is NOT built here, the 3x3 algebra from F on is a host helper
 * (fsgm_epipolar_from_F below) -- and a dense half: the
 * per-pixel maps Pd0 / normlizeDirection / Offset / Rflow from F, H, the epipole and the direction
 * flag (:99-115, rotation_motion.m).  The dense half and the tail of epipolar_sgm_of.m (:33-51: gray
 * conversion, calc_cost_sgm, flow = disparity * direction + rotation flow) run on the device, so a
 * call uploads the image pair and 21 numbers instead of 18.6 MB of fp64 maps per 1242x375 frame.
 * Matrices are row-major (F[3*i+j] = F(i+1,j+1)); pixel coordinates as in MATLAB (1-based outputs).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    double F[9];          /* fundamental matrix                                   (epipolar_geometry.m:30) */
    double H[9];          /* rotation-compensating homography K*R/K               (:62) */
    double epipole[2];    /* epipole in image 2, epi(1:2)                          (:43-45) */
    int32_t direction;    /* 0 = expansion, 1 = contraction (directions negated)  (:92-96,:108-110) */
} fsgm_epi_geometry;

/* The in-tree remainder of the sparse half: epipolar_geometry.m:40-96 -- from the fundamental matrix F (however it was estimated:
 * the SURF + LMedS front end, :130-149, is MATLAB toolbox code and not built), the intrinsics K and the inlier matches to the
 * epipole in image 2 (svd(F'), :40-43), H = K*R/K with R the rotation of E = K'FK that is close to the identity (:46-65) and the
 * expansion / contraction vote (:68-96).  3x3 host algebra, no GPU.  pts1 / pts2: n_points x (x, y) in MATLAB's 1-based pixel
 * coordinates; inliers (may be NULL = all) one byte per match.  *ambiguous (may be NULL) = 1 when the diagonal test of :58-62
 * accepts both or neither of the two candidate rotations (the reference's answer then depends on its SVD's signs).  UNPINNED:
 * MATLAB's svd is LAPACK's, this is a Jacobi solver; the quantities used are invariant to an SVD's sign freedoms. */
fsgm_status fsgm_epipolar_from_F(const double* F /* 9, row-major */, const double* K /* 9 */, int32_t n_points, const double* pts1,
                                 const double* pts2, const uint8_t* inliers, fsgm_epi_geometry* out, int32_t* ambiguous);

/* [PrefD0, NormlizeDirection, Offset, Rflow] of epipolar_geometry.m:99-115: f64 [2][H][W] / [H][W] */
fsgm_status fsgm_epipolar_maps_host(const fsgm_epi_geometry* g, int32_t width, int32_t height, double* Pd0,
                                    double* normDirect, double* Offset, double* Rflow, int32_t device);
/* [flow, minC] = epipolar_sgm_of(I0, I1, K, dMax, vMax) from :33 on, the geometry given.  Images u8
 * [channels][H][W] (1 or 3 planes, x fastest); flow f64 [3][H][W] (third plane 1, :51); minC (may be
 * NULL) u32 [H][W].  P1 = 6, P2 = 64 (:19).  prm (may be NULL): paths etc. as for calc_cost_sgm. */
fsgm_status fsgm_epipolar_sgm_of_host(const uint8_t* I0, const uint8_t* I1, int32_t width, int32_t height,
                                      int32_t channels, const fsgm_epi_geometry* g, int32_t dMax, double vMax,
                                      const fsgm_epi_params* prm, double* flow, uint32_t* minC);

/* ------------------------------------------------------------------------------------------
 * calc_pyd_cost_sgm  (calc_pyd_cost_sgm.cpp:439-510, called from pyramidal_sgm.m:50)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t* I1;            /* prhs[0]  u8 [H][W] */
    const uint8_t* I2;            /* prhs[1] */
    int32_t width, height;        /* mxGetM / mxGetN of prhs[0] (calc_pyd_cost_sgm.cpp:454-455) */
    const double* preMv;          /* prhs[2]  f64 [2][mvHeight][mvWidth], plane 0 = x; indexed with its own stride */
    int32_t mvWidth, mvHeight;    /* mxGetM(prhs[2]), mxGetN(prhs[2])/2 (:493-494); must be >= width, height */
    int32_t halfSearchWinSizeX;   /* prhs[3] */
    int32_t halfSearchWinSizeY;   /* prhs[4] */
    int32_t aggHalfWinSize;       /* prhs[5] */
    int32_t subPixelRefine;       /* prhs[6] */
    int32_t P1, P2;               /* prhs[7], prhs[8] */
    int32_t enableDiagnalPath;    /* prhs[9]  (bool) */
    int32_t totalPass;            /* prhs[10] */
    int32_t adpativeP2;           /* prhs[11] (bool) */
} fsgm_pyd_in;

typedef struct {
    uint32_t* bestD;              /* plhs[0]  u32 [H][W], 0-based candidate index sx*Sy+sy */
    uint32_t* minC;               /* plhs[1]  u32 [H][W] */
    double*   mvSub;              /* plhs[2]  f64 [2][H][W], plane 0 = x; zeros when subPixelRefine == 0 */
    uint8_t*  C;                  /* optional debug tap: cost volume u8 [H][W][D] (calc_pyd_cost_sgm.cpp:496) */
    uint32_t* S;                  /* optional debug tap: summed path costs u32 [H][W][D] (:125) */
} fsgm_pyd_out;

fsgm_status fsgm_calc_pyd_cost_sgm_host(const fsgm_pyd_in* in, const fsgm_pyd_out* out, int32_t device);
fsgm_status fsgm_calc_pyd_cost_sgm_batch_host(int32_t n_frames, const fsgm_pyd_in* in,
                                              const fsgm_pyd_out* out, int32_t device);

typedef struct fsgm_pyd_plan fsgm_pyd_plan;
fsgm_status fsgm_pyd_plan_create(fsgm_pyd_plan** plan, int32_t width, int32_t height,
                                 int32_t mvWidth, int32_t mvHeight, int32_t halfSearchWinSizeX,
                                 int32_t halfSearchWinSizeY, int32_t aggHalfWinSize, int32_t batch,
                                 int32_t device);
void        fsgm_pyd_plan_destroy(fsgm_pyd_plan* plan);
fsgm_status fsgm_pyd_plan_set_params(fsgm_pyd_plan* plan, int32_t P1, int32_t P2,
                                     int32_t enableDiagnalPath, int32_t totalPass,
                                     int32_t adpativeP2, int32_t subPixelRefine);
fsgm_status fsgm_pyd_plan_upload(fsgm_pyd_plan* plan, int32_t frame, const uint8_t* I1,
                                 const uint8_t* I2, const double* preMv);
fsgm_status fsgm_pyd_plan_upload_cost(fsgm_pyd_plan* plan, int32_t frame, const uint8_t* C);
fsgm_status fsgm_pyd_plan_run(fsgm_pyd_plan* plan, int32_t stages);   /* FSGM_STAGE_* bits */
fsgm_status fsgm_pyd_plan_download(fsgm_pyd_plan* plan, int32_t frame, uint32_t* bestD,
                                   uint32_t* minC, double* mvSub);
fsgm_status fsgm_pyd_plan_download_cost(fsgm_pyd_plan* plan, int32_t frame, uint8_t* C);
fsgm_status fsgm_pyd_plan_download_sum(fsgm_pyd_plan* plan, int32_t frame, uint32_t* S);
fsgm_status fsgm_pyd_plan_time(fsgm_pyd_plan* plan, int32_t stages, int32_t warmup,
                               int32_t iters, float* ms_avg);

/* ------------------------------------------------------------------------------------------
 * pyramidal_sgm  (pyramidal_sgm.m:1-77 -- the MATLAB driver around calc_pyd_cost_sgm)
 *
 * The whole level loop on the device: image pyramid (impyramid 'reduce', :28-31), rgb2gray (:44-45),
 * one calc_pyd_cost_sgm per level from coarse to fine (:50), index -> motion vector + previous
 * level's hint + sub-pixel part (:57-64), and 2*imresize(mv, 2, 'nearest') as the next level's hint
 * map (:72).  One upload of the image pair, one download of the flow; the reference's loop crosses
 * the MEX boundary once per level.
 *
 * Images: u8 [channels][height][width], channels = 1 (gray) or 3 (R, G, B planes), x fastest -- what
 * permute(I, [2 1 3]) hands a MEX (pyramidal_sgm.m:44).  Flows: f64 [2][height][width], plane 0 = x.
 * impyramid / rgb2gray / imresize are MATLAB toolbox functions that are not in the reference tree;
 * they follow the published behaviour restated in oracle/fsgm_oracle_pyramid.cpp.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t numPyd;                 /* pyramid levels          (pyramidal_sgm.m:12, default 5; test_psgm.m:33 passes 3) */
    int32_t P1, P2;                 /* penalties               (:15-16: 6, 32) */
    int32_t aggHalfWinSize;         /* cost aggregation window (:17: 2) */
    int32_t verSearchHalfWinSize;   /* search window, y        (:18: 5) */
    int32_t horSearchHalfWinSize;   /* search window, x        (:19: 5) */
    int32_t enableDiagonal;         /* :20: 1 */
    int32_t totalPass;              /* :21: 2 */
    int32_t adaptiveP2;             /* :22: 0 */
    int32_t device;                 /* HIP device ordinal */
} fsgm_pyramid_params;

fsgm_pyramid_params fsgm_pyramid_params_default(void);

/* [mvCurLevel, mvPyd, minC] = pyramidal_sgm(I0, I1, numPyd).  mv: level-1 flow f64 [2][H][W]; minC
 * (may be NULL): level-1 minimum summed path cost u32 [H][W]; mvPyd (may be NULL): numPyd pointers,
 * entry l-1 (if not NULL) receives level l's flow f64 [2][H_l][W_l], H_l = ceil(H_{l-1}/2). */
fsgm_status fsgm_pyramidal_sgm_host(const uint8_t* I0, const uint8_t* I1, int32_t width, int32_t height,
                                    int32_t channels, const fsgm_pyramid_params* prm,
                                    double* mv, uint32_t* minC, double* const* mvPyd);

typedef struct fsgm_pyramid_plan fsgm_pyramid_plan;
fsgm_status fsgm_pyramid_plan_create(fsgm_pyramid_plan** plan, int32_t width, int32_t height,
                                     int32_t channels, const fsgm_pyramid_params* prm);
/* the same with `batch` image pairs resident: every kernel of a level covers all of them (frames are independent; a level
 * still needs the level above).  upload_frame / download_frame address one pair; upload / download are frame 0. */
fsgm_status fsgm_pyramid_plan_create_batch(fsgm_pyramid_plan** plan, int32_t width, int32_t height, int32_t channels,
                                           const fsgm_pyramid_params* params, int32_t batch);
fsgm_status fsgm_pyramid_plan_upload_frame(fsgm_pyramid_plan* plan, int32_t frame, const uint8_t* I0, const uint8_t* I1);
fsgm_status fsgm_pyramid_plan_download_frame(fsgm_pyramid_plan* plan, int32_t frame, int32_t level, double* mv, uint32_t* minC);
void        fsgm_pyramid_plan_destroy(fsgm_pyramid_plan* plan);
/* size of pyramid level `level` (1 = full resolution) */
fsgm_status fsgm_pyramid_plan_level_size(fsgm_pyramid_plan* plan, int32_t level, int32_t* width, int32_t* height);
fsgm_status fsgm_pyramid_plan_upload(fsgm_pyramid_plan* plan, const uint8_t* I0, const uint8_t* I1);
fsgm_status fsgm_pyramid_plan_run(fsgm_pyramid_plan* plan);              /* asynchronous */
/* wait for everything queued on the plan.  Plans are independent (own buffers, own stream): several of them
 * run concurrently when each is started with _run before any is waited for -- the throughput form of the
 * driver loop (pyramidal_sgm.m:37-75 handles one pair at a time) */
fsgm_status fsgm_pyramid_plan_sync(fsgm_pyramid_plan* plan);
/* only the image half of the loop (impyramid, rgb2gray: pyramidal_sgm.m:28-31, 44-45), for callers that run
 * another matcher per level (fsgm_amd.pyramidal_sgm_ng swaps calc_pyd_cost_sgm_ng in); asynchronous */
fsgm_status fsgm_pyramid_plan_run_images(fsgm_pyramid_plan* plan);
/* HBM -> host (synchronous): flow and/or minC of one level; either pointer may be NULL */
fsgm_status fsgm_pyramid_plan_download(fsgm_pyramid_plan* plan, int32_t level, double* mv, uint32_t* minC);
/* debug tap: the gray images calc_pyd_cost_sgm saw at one level (after impyramid and rgb2gray), u8 [H_l][W_l] */
fsgm_status fsgm_pyramid_plan_download_gray(fsgm_pyramid_plan* plan, int32_t level, uint8_t* gray0, uint8_t* gray1);   /* frame 0 */
fsgm_status fsgm_pyramid_plan_download_gray_frame(fsgm_pyramid_plan* plan, int32_t frame, int32_t level, uint8_t* gray0, uint8_t* gray1);
/* average milliseconds of one whole pyramid run (HIP events on the plan's stream) */
fsgm_status fsgm_pyramid_plan_time(fsgm_pyramid_plan* plan, int32_t warmup, int32_t iters, float* ms_avg);

/* ---- the pyramidal level loop with the neighbour-guided matcher ----
 * BASELINE config 4 names calc_pyd_cost_sgm_ng for the pyramidal path.  The reference ships that MEX as a swap-in
 * (same 8 arguments as ng_sgm.m:20) without a driver; this is pyramidal_sgm.m's loop (:24-76) around it: level
 * images by impyramid 'reduce' and rgb2gray (:28-31, :44-45), the coarsest level from a zero hint map (:34), every
 * finer level from 2*imresize(flow, 2, 'nearest') of the level above (:72) -- the MEX returns the flow itself
 * (calc_pyd_cost_sgm_ng.cpp:296-297), so no index-to-vector step.  Everything stays in HBM between levels. */
typedef struct {
    int32_t numPyd;                 /* levels, 1..16 */
    int32_t P1, P2;                 /* ng_sgm.m:7-8: 6, 32 */
    int32_t halfSearchWinSize;      /* ng_sgm.m:20: 1  -> 81 candidates per pixel */
    int32_t aggSize;                /* ng_sgm.m:20: 2 */
    int32_t subPixelRefine;         /* ng_sgm.m:20: 0 */
    int32_t device;
} fsgm_ng_pyramid_params;
fsgm_ng_pyramid_params fsgm_ng_pyramid_params_default(void);
typedef struct fsgm_ng_pyramid_plan fsgm_ng_pyramid_plan;
fsgm_status fsgm_ng_pyramid_plan_create(fsgm_ng_pyramid_plan** plan, int32_t width, int32_t height, int32_t channels,
                                        const fsgm_ng_pyramid_params* prm);
/* the same with `batch` image pairs resident: one run takes all of them through every level together (frames are
 * independent; every kernel of a level covers the whole batch) */
fsgm_status fsgm_ng_pyramid_plan_create_batch(fsgm_ng_pyramid_plan** plan, int32_t width, int32_t height, int32_t channels,
                                              int32_t batch, const fsgm_ng_pyramid_params* prm);
fsgm_status fsgm_ng_pyramid_plan_upload_frame(fsgm_ng_pyramid_plan* plan, int32_t frame, const uint8_t* I0, const uint8_t* I1);
fsgm_status fsgm_ng_pyramid_plan_download_frame(fsgm_ng_pyramid_plan* plan, int32_t frame, int32_t level, double* flow, uint32_t* minC);
void        fsgm_ng_pyramid_plan_destroy(fsgm_ng_pyramid_plan* plan);
fsgm_status fsgm_ng_pyramid_plan_level_size(fsgm_ng_pyramid_plan* plan, int32_t level, int32_t* width, int32_t* height);
fsgm_status fsgm_ng_pyramid_plan_upload(fsgm_ng_pyramid_plan* plan, const uint8_t* I0, const uint8_t* I1);
fsgm_status fsgm_ng_pyramid_plan_run(fsgm_ng_pyramid_plan* plan);        /* asynchronous */
/* flow [2][h][w] and minC [h][w] of `level` (1 = full resolution); either may be NULL */
fsgm_status fsgm_ng_pyramid_plan_download(fsgm_ng_pyramid_plan* plan, int32_t level, double* flow, uint32_t* minC);
fsgm_status fsgm_ng_pyramid_plan_time(fsgm_ng_pyramid_plan* plan, int32_t warmup, int32_t iters, float* ms_avg);
/* one call = the whole loop on a plan cached per shape and parameters: I0, I1 as for fsgm_pyramidal_sgm_host;
 * flow [2][height][width] and minC (may be NULL) of level 1; flowPyd (may be NULL): numPyd pointers, entry l-1
 * receives the flow of level l ([2][h_l][w_l]) or is skipped when NULL */
fsgm_status fsgm_pyramidal_sgm_ng_host(const uint8_t* I0, const uint8_t* I1, int32_t width, int32_t height,
                                       int32_t channels, const fsgm_ng_pyramid_params* prm,
                                       double* flow, uint32_t* minC, double* const* flowPyd);


/* ------------------------------------------------------------------------------------------
 * Post-processing  (SURVEY 8(f) N3): the MATLAB functions the evaluation script chains after SGM
 * (test.m:45-50) -- speckle_filter.m, calc_disp_from_first.m, forward_backward_check.m,
 * scanline_in_fill.m, vzInd2Disp.m -- one entry point per function, same argument meaning.
 *
 * Maps are f64 [height][width], x fastest; NaN = invalid (MATLAB's NaN).  Pd0 / normDirect are
 * f64 [2][height][width] with plane 0 = x, Pd0 in MATLAB's 1-based pixel coordinates; O is
 * offsetFromPosD0; n = dMax + 1 (test.m:6).  (These are the epipolar maps of calc_cost_sgm above.)
 * ------------------------------------------------------------------------------------------ */
/* speckle_filter.m:1 [imageFiltered, labelImage] = speckle_filter(image, maxDiff, maxSpeckleSize): every
 * 4-connected region (neighbours joined when both valid and |a-b| < maxDiff) of fewer than
 * maxSpeckleSize pixels becomes NaN.  labelImage (may be NULL): i32 [H][W], regions numbered in raster
 * order of their first pixel, 0 where the input is NaN.  Defaults of the original: 2, 100. */
fsgm_status fsgm_speckle_filter_host(const double* image, int32_t width, int32_t height, double maxDiff,
                                     double maxSpeckleSize, double* imageFiltered, int32_t* labelImage,
                                     int32_t device);
/* calc_disp_from_first.m:1 D2 = calc_disp_from_first(D1, Pd0, normDirect, O, vMax, n): -1 where no pixel
 * of D1 lands.  D1 must hold non-negative values or NaN (vz indices are): checked. */
fsgm_status fsgm_calc_disp_from_first_host(const double* D1, int32_t width, int32_t height, const double* Pd0,
                                           const double* normDirect, const double* O, double vMax, double n,
                                           double* D2, int32_t device);
/* forward_backward_check.m:1 D1 = forward_backward_check(D1, D2, Pd0, normDirect, O, vMax, n), threshold 2.0 (:6) */
fsgm_status fsgm_forward_backward_check_host(const double* D1, const double* D2, int32_t width, int32_t height,
                                             const double* Pd0, const double* normDirect, const double* O,
                                             double vMax, double n, double* D1checked, int32_t device);
/* scanline_in_fill.m:2 output = scanline_in_fill(input), one channel (test.m:49 passes a 2-D map) */
fsgm_status fsgm_scanline_in_fill_host(const double* input, int32_t width, int32_t height, double* output,
                                       int32_t device);
/* vzInd2Disp.m:1 D = vzInd2Disp(w, O, vMax, n) */
fsgm_status fsgm_vzind2disp_host(const double* w, const double* O, int32_t width, int32_t height, double vMax,
                                 double n, double* D, int32_t device);
/* vmf.m:1 flowMed = vmf(flow): medfilt2(flow(:,:,c), [5 5]) per channel (zero padding, 13th smallest of 25);
 * flow f64 [channels][H][W], 1..3 channels (test.m:76 passes the 3-plane flow) */
fsgm_status fsgm_vmf_host(const double* flow, int32_t width, int32_t height, int32_t channels, double* flowMed,
                          int32_t device);
/* test.m:45-50 in one call, intermediates resident in HBM:
 *   filterD1 = speckle_filter(D1, 2, 100); filterD2 = calc_disp_from_first(filterD1, ...);
 *   filterD1 = forward_backward_check(filterD1, filterD2, ...); filterD1 = speckle_filter(filterD1, dMax, rows*cols/10);
 *   filterD1 = scanline_in_fill(filterD1); disp = vzInd2Disp(filterD1, O, vMax, n)
 * filterD2 and disp may be NULL. */
fsgm_status fsgm_epi_postprocess_host(const double* D1, int32_t width, int32_t height, const double* Pd0,
                                      const double* normDirect, const double* O, double vMax, double n,
                                      double dMax, double* filterD1, double* filterD2, double* disp,
                                      int32_t device);

typedef struct fsgm_post_plan fsgm_post_plan;
fsgm_status fsgm_post_plan_create(fsgm_post_plan** plan, int32_t width, int32_t height, int32_t device);
void        fsgm_post_plan_destroy(fsgm_post_plan* plan);
/* host -> HBM; any pointer may be NULL to keep what the plan holds */
fsgm_status fsgm_post_plan_upload(fsgm_post_plan* plan, const double* D1, const double* Pd0,
                                  const double* normDirect, const double* O);
fsgm_status fsgm_post_plan_run(fsgm_post_plan* plan, double vMax, double n, double dMax);   /* test.m:45-50, asynchronous */
fsgm_status fsgm_post_plan_download(fsgm_post_plan* plan, double* filterD1, double* filterD2, double* disp);
fsgm_status fsgm_post_plan_time(fsgm_post_plan* plan, double vMax, double n, double dMax, int32_t warmup,
                                int32_t iters, float* ms_avg);

/* ------------------------------------------------------------------------------------------
 * calc_pyd_cost_sgm_ng  (calc_pyd_cost_sgm_ng.cpp:448-523; same 8-argument list as the call in
 * ng_sgm.m:20, no caller in the reference tree)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t* I1;            /* prhs[0] */
    const uint8_t* I2;            /* prhs[1] */
    int32_t width, height;
    const double* preMv;          /* prhs[2]  f64 [2][mvHeight][mvWidth]; hints are clamped to the map (:392-393) */
    int32_t mvWidth, mvHeight;
    int32_t halfSearchWinSize;    /* prhs[3] */
    int32_t aggSize;              /* prhs[4]  aggregation radius = (int)aggSize/2 (:490) */
    int32_t subPixelRefine;       /* prhs[5] */
    int32_t P1, P2;               /* prhs[6], prhs[7] */
} fsgm_ng_in;

typedef struct {
    uint32_t* minC;               /* plhs[0]  u32 [H][W] */
    double*   flow;               /* plhs[1]  f64 [2][H][W], plane 0 = x */
    uint32_t* S;                  /* optional debug tap: summed path costs u32 [H][W][D], D = 9*(2r+1)^2 */
} fsgm_ng_out;

fsgm_status fsgm_calc_pyd_cost_sgm_ng_host(const fsgm_ng_in* in, const fsgm_ng_out* out, int32_t device);
fsgm_status fsgm_calc_pyd_cost_sgm_ng_batch_host(int32_t n_frames, const fsgm_ng_in* in,
                                                 const fsgm_ng_out* out, int32_t device);

/* ------------------------------------------------------------------------------------------
 * calc_cost_sgm_ng  (calc_cost_sgm_ng.cpp:484-526, called from ng_sgm.m:20; prhs[2..5] ignored)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t* I1;            /* prhs[0] */
    const uint8_t* I2;            /* prhs[1] */
    int32_t width, height;
    int32_t P1, P2;               /* prhs[6], prhs[7] */
    /* The reference draws 2 x rand() per random hint in raster order (:148-149): 8 per pixel.
     * rand_stream = those draws (fsgm_sgm_ng_rand_draws(width,height) values); NULL makes the
     * library call libc rand() itself, which is what the reference does (process-global state). */
    const int32_t* rand_stream;
} fsgm_otf_in;

typedef struct {
    uint32_t* minC;               /* plhs[0]  u32 [H][W] */
    double*   flow;               /* plhs[1]  f64 [2][H][W] */
} fsgm_otf_out;

int64_t     fsgm_sgm_ng_rand_draws(int32_t width, int32_t height);
fsgm_status fsgm_calc_cost_sgm_ng_host(const fsgm_otf_in* in, const fsgm_otf_out* out, int32_t device);
fsgm_status fsgm_calc_cost_sgm_ng_batch_host(int32_t n_frames, const fsgm_otf_in* in,
                                             const fsgm_otf_out* out, int32_t device);

/* ------------------------------------------------------------------------------------------
 * Device lists: a batch over several GPUs of one node from ONE process  (SURVEY 8(b) "batch variants", 8(e);
 * north_star: the host stays MATLAB -- one process -- and a batch of frames shards across the 8 GPUs of a node).
 *
 * Frame i of the batch runs on devices[i % n_devices]; every list entry gets a host thread of its own that runs its frames
 * through the single-device batch call above (own cached plan, own stream); nothing is exchanged between devices -- frames
 * are independent (calc_cost_sgm.cpp has no state across calls), so there is no collective and no peer copy.  Entries are HIP
 * ordinals taken modulo the number of devices present: {0,1,...,7} is valid on any box, on a one-GPU box all entries share the
 * device and take turns.  Results are identical to single calls whatever the list.  Errors: the first failing entry's status
 * and message, prefixed with the entry.  The MEX gateway calc_cost_sgm takes its list from FSGM_DEVICES=0,1,... when it is
 * handed a batch.
 * ------------------------------------------------------------------------------------------ */
fsgm_status fsgm_calc_cost_sgm_batch_devices_host(int32_t n_frames, const fsgm_epi_in* in, const fsgm_epi_out* out,
                                                  const fsgm_epi_params* prm /* device field ignored */,
                                                  int32_t n_devices, const int32_t* devices);
fsgm_status fsgm_calc_pyd_cost_sgm_batch_devices_host(int32_t n_frames, const fsgm_pyd_in* in, const fsgm_pyd_out* out,
                                                      int32_t n_devices, const int32_t* devices);
fsgm_status fsgm_calc_pyd_cost_sgm_ng_batch_devices_host(int32_t n_frames, const fsgm_ng_in* in, const fsgm_ng_out* out,
                                                         int32_t n_devices, const int32_t* devices);
/* frames whose rand_stream is NULL get their libc rand() draws on the calling thread, in frame order, before they scatter:
 * the draws a sequence of single calls would have made (calc_cost_sgm_ng.cpp:148-149) */
fsgm_status fsgm_calc_cost_sgm_ng_batch_devices_host(int32_t n_frames, const fsgm_otf_in* in, const fsgm_otf_out* out,
                                                     int32_t n_devices, const int32_t* devices);
/* image pairs through the pyramidal drivers: arguments per pair as fsgm_pyramidal_sgm_host / fsgm_pyramidal_sgm_ng_host take them */
typedef struct {
    const uint8_t* I0;
    const uint8_t* I1;
    double*   mv;                 /* level-1 flow f64 [2][H][W] */
    uint32_t* minC;               /* may be NULL */
    double* const* mvPyd;         /* may be NULL: numPyd pointers, per-level flows */
} fsgm_pyramid_pair;
fsgm_status fsgm_pyramidal_sgm_batch_devices_host(int32_t n_pairs, const fsgm_pyramid_pair* pairs, int32_t width, int32_t height,
                                                  int32_t channels, const fsgm_pyramid_params* prm /* device field ignored */,
                                                  int32_t n_devices, const int32_t* devices);
fsgm_status fsgm_pyramidal_sgm_ng_batch_devices_host(int32_t n_pairs, const fsgm_pyramid_pair* pairs, int32_t width, int32_t height,
                                                     int32_t channels, const fsgm_ng_pyramid_params* prm /* device field ignored */,
                                                     int32_t n_devices, const int32_t* devices);
/* "0,1,2" (commas, semicolons or blanks) -> devices[]; returns the number of entries, 0 for an empty / NULL text, -1 for anything
 * that is not a list of non-negative integers */
int32_t fsgm_parse_device_list(const char* text, int32_t* devices, int32_t max_devices);
/* the partition itself, for callers that want to know it: the frames of entry `slot` of a list of n_devices entries
 * (frames may be NULL to get the count only) */
void fsgm_shard_frames(int32_t n_frames, int32_t n_devices, int32_t slot, int32_t* frames, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* FSGM_H */
