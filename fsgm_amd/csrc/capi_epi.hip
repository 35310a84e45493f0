// capi_epi.hip -- C ABI for the calc_cost_sgm path: device-resident plan + host-pointer entry
// points (what the calc_cost_sgm mexFunction gateway calls).  See include/fsgm.h.
#include "capi_common.h"
#include "epi_kernels.h"
#include "geometry_kernels.h"
#include "pyramid_kernels.h"
#include <algorithm>
#include <cmath>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace fsgm {
char* last_error_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace fsgm

using namespace fsgm;

struct fsgm_epi_plan {
    int W = 0, H = 0, D = 0, batch = 0;
    fsgm_epi_params prm{};
    int P1 = 6, P2 = 64;                 // epipolar_sgm_of.m:19
    double vMax = 0.3;                   // epipolar_sgm_of.m:16
    size_t NP = 0, N = 0;                // pixels, voxels per frame
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t *dI1 = nullptr, *dI2 = nullptr;
    uint32_t *dCen1 = nullptr, *dCen2 = nullptr;
    double *dPd0 = nullptr, *dNd = nullptr, *dOff = nullptr, *dVz = nullptr;
    double vzmax = 0.0;         // max |vzInd(d)| of the table in dVz (inf when not finite)
    uint8_t *dCraw = nullptr, *dC = nullptr, *dL = nullptr;
    uint32_t *dBestD = nullptr, *dMinC = nullptr, *dS = nullptr;
    uint32_t *dD2enc = nullptr, *dD2 = nullptr;          // forward-backward check (prm.fb_check)
    uint8_t* dConf = nullptr;
    // fused-sweep aggregation (epi_sweep.hip): horizontal path costs, u16 sums, block-boundary states
    // (see enqueue(): horizontal kernel on stream_h; the frames split into two lanes that sweep
    // down then up on stream / stream_b)
    uint8_t *dLh = nullptr, *dX = nullptr, *dXup = nullptr, *dState = nullptr, *dCkpt = nullptr, *dCkptV = nullptr;
    // parallel sweeps (sweep_par): Y_up of every frame and the up sweep's own block-boundary states
    uint8_t *dXupAll = nullptr, *dStateUp = nullptr;
    uint8_t* dLx = nullptr;              // parallel sweeps of few frames: the two along-x path volumes [batch][2][N] (par_x_lines)
    bool sweep_par = false;              // AGG_SWEEP only: down and up sweeps side by side, WTA over the three Y volumes
    bool sweep_mid = false;              // sweep_par only: the two sweeps meet in the middle, each finishing the other's half with the WTA inside
    // band sweeps (epi_band.hip): the first pass's 9th bits, the hand-off between the bands of a frame; dX, dRec, dS0 as above
    uint32_t* dBits = nullptr;
    uint4* dBandEdge = nullptr;
    size_t band_edge_maps = 0;           // hand-off maps per frame the buffer holds (1: bands in sequence; bands - 1: chained)
    uint32_t *dBandTicket = nullptr, *dBandErr = nullptr;   // chained band sweeps: work counter, give-up flag of the bounded waits
    uint32_t band_salt = 0;              // launch sequence number of the chained form (hand-off tags)
    bool band_chain = false;             // AGG_BAND only: one workgroup per (band, frame) instead of one per frame
    bool band_edge_untagged = false;     // the hand-off maps hold words without this scheme's tags (sequential form, S tap): refill before a chained launch
    // epipolar driver (fsgm_epipolar_sgm_of_host): rotation flow, composed flow, RGB staging
    double *dRflow = nullptr, *dFlow = nullptr;
    uint8_t* dRgb = nullptr;
    uint2* dRec = nullptr;
    uint16_t* dS0 = nullptr;
    size_t state_stride = 0;
    hipStream_t stream_h = nullptr, stream_b = nullptr, stream_c = nullptr;
    hipEvent_t ev_fork = nullptr, ev_h = nullptr, ev_b = nullptr, ev_c = nullptr;
    hipEvent_t ev_hl[3] = {nullptr, nullptr, nullptr};   // the horizontal pair of each frame lane done (sweep pipeline)
    bool pair_split = true;              // FSGM_EPI_PAIRSPLIT=0: one pair launch for all lanes (A/B switch)
    int lanes = 2;                       // frame lanes of the sweeps (FSGM_EPI_LANES, 1..3)
    std::vector<int> cmax;               // per frame: upper bound of the cost values in dC
    bool vz_valid = false;
    int agg_mode = 0;                    // 0 auto, 1 per-direction line kernels, 2 fused sweeps (if eligible), 3 parallel sweeps, 4 / 5 band sweeps, 6 sweeps meeting in the middle
    int cus = 256;                       // compute units of the device (band sweeps: one workgroup per frame, two per CU)
    int kernel_kind = AGG_GENERIC;
    bool packed = false;
};

// batch sizes at which auto mode moves from the line kernels to the parallel sweeps and on to the full sweep pipeline
// (8 paths; measured at 1242x375x128); FSGM_EPI_PAR_MIN / FSGM_EPI_PAR_MAX override them
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
// The switch points were measured at 1242x375x128.  A fused pipeline's fixed latency goes with a linear dimension of the
// frame and the line kernels' time per frame with its voxels, so the batch at which the two cross goes with
// voxels^(-2/3): 320x240x64 (BASELINE configs[1], 1/12 of the voxels) measured ~26 / ~80 / ~48 frames for the three switches
// against 4 / 18 / 9 at the KITTI shape (profiles/r03_crossover_320x240x64.txt); the scaled values are 21 / 95 / 47.
// An environment override is taken as it stands.  FSGM_EPI_SHAPE_SCALE=0: the KITTI values for every shape.
static int scaled_batch(int W, int H, int D, int at_kitti) {
    static const int on = env_int("FSGM_EPI_SHAPE_SCALE", 1);
    if (!on) return at_kitti;
    const double s = std::pow(1242.0 * 375.0 * 128.0 / ((double)W * H * D), 2.0 / 3.0);
    return std::max(1, (int)std::lround(at_kitti * s));
}
static int switch_batch(const char* env_name, int W, int H, int D, int at_kitti) {
    const int e = env_int(env_name, -1);
    return e >= 0 ? e : scaled_batch(W, H, D, at_kitti);
}
static int par_min_batch(int W, int H, int D) { return switch_batch("FSGM_EPI_PAR_MIN", W, H, D, 4); }
static int par_max_batch(int W, int H, int D) { return switch_batch("FSGM_EPI_PAR_MAX", W, H, D, 26); }
// from this batch on the parallel sweeps meet in the middle (mode 6; profiles/r04_sweep_mid.txt: 10 frames 1.38 vs 1.44 ms, 18 frames
// 2.06 vs 2.41 -- and 2.61 for the full pipeline --, 28 frames 3.49 vs 3.35 for the full pipeline); FSGM_EPI_MID_MIN=0: never
static int mid_min_batch(int W, int H, int D) { return switch_batch("FSGM_EPI_MID_MIN", W, H, D, 10); }
// Band sweeps (all four paths of a pass in one sweep, one workgroup per frame, two workgroups per CU) in auto mode: a launch
// takes as long as its slowest CU -- measured at 1242x375x128, 8 paths, 256 CUs: 25.2 ms with one workgroup per CU (up to
// 256 frames), 42.8 ms with two (up to 512) -- while the block sweeps take 0.107 ms per frame whatever the count.  In units of
// the block sweeps' time per frame a round of the band kernel costs 0.92 x CUs (half filled) or 0.785 x 2 CUs (full): auto
// mode takes the band sweeps where that is less than the batch (256 frames, 402..512, 638..768, ...).
// FSGM_EPI_BAND_MIN: never below this many frames (0: never at all).
static int band_min_batch() { static const int v = env_int("FSGM_EPI_BAND_MIN", 64); return v; }
// (4 paths, against the pair pipeline's 0.084 ms per frame: 17.0 / 29.0 ms per round -> 0.74 x CUs / 0.63 x 2 CUs)
// The chained form (one workgroup per band and frame, mode 5) has no rounds: measured 7 ms + 0.076 ms per frame at 8 paths
// (profiles/r03_band_chain.txt: 96 .. 512 frames) = 0.26 x CUs + 0.71 per frame in the same units (4 paths: 0.23 x CUs + 0.68);
// it takes the batches between the sequential form's rounds (257 .. ~470 frames, 513 .. ~700, ...).
// Returns 0: neither pays, 1: sequential band sweeps, 2: chained.
// Other shapes: a band workgroup walks nbands x (W + skew x R) steps of R rows for H x W pixels (skew 2 with the diagonals,
// 1 without), the block sweeps' time goes with H x W: the costs above are scaled by that ratio relative to the KITTI shape
// (6 bands of 64 rows, 1242 columns).  320x240x64: a round measured 1.18 x 2 CUs (8 paths) and 0.90 x 2 CUs (4 paths) against
// 0.785 / 0.63 at the KITTI shape; the ratio gives 1.33 / 0.88 -- the band sweeps never pay there at 8 paths, from 512 frames at 4.
static double band_shape_cost(int W, int H, int D, int paths) {
    const int skew = paths == 8 ? 2 : 1;
    auto eff = [&](double w, double h, int R) { const int nb = ((int)h + R - 1) / R; return (h / (nb * R)) * (w / (w + skew * R)); };
    return eff(1242.0, 375.0, 64) / eff((double)W, (double)H, band_rows(D));
}
static int band_choice(int batch, int cus, int paths, int W, int H, int D) {
    if (band_min_batch() <= 0 || batch < band_min_batch()) return 0;
    const double g = band_shape_cost(W, H, D, paths);
    const double half = g * (paths == 8 ? 0.92 : 0.74), whole = g * (paths == 8 ? 0.785 : 0.63);
    const int slots = 2 * cus, full = batch / slots, tail = batch % slots;
    const double seq = full * whole * slots + (tail == 0 ? 0.0 : (tail <= cus ? half * cus : whole * slots));
    static const int chain_ok = env_int("FSGM_EPI_BAND_CHAIN", 1);                      // 0: auto mode never takes the chained form
    const double chain = chain_ok ? g * ((paths == 8 ? 0.26 : 0.23) * cus + (paths == 8 ? 0.71 : 0.68) * batch) : 1e30;
    if (std::min(seq, chain) >= (double)batch) return 0;
    return seq <= chain ? 1 : 2;
}
static int pairs_min_batch(int W, int H, int D) { return switch_batch("FSGM_EPI_PAIRS_MIN", W, H, D, 9); }   // 4 paths: line kernels -> pair pipeline

// What runs for a plan of this shape, batch and parameter set (cm: the largest cost in the volumes): a function of its
// arguments and the FSGM_EPI_* environment only, so that fsgm_epi_auto_pipeline can answer without a plan.
struct PipelineChoice { int kind; bool sweep_par, band_chain, sweep_mid; };
static PipelineChoice choose_pipeline(int W, int H, int D, int batch, int paths, int P1, int P2, int cm, int agg_mode, int cus) {
    PipelineChoice c = {AGG_GENERIC, false, false, false};
    if (agg_packed_lpp(D) == 0) return c;
    const bool nowrap = P1 >= 0 && P2 >= 0 && cm + P2 + std::max(P1, P2) <= 255;
    c.kind = nowrap ? AGG_PACKED_NOWRAP : AGG_PACKED_WRAP;
    // the fused sweeps cover the 8-path no-wrap case; everything else stays on the line kernels
    // Auto mode takes the fused pipelines only for batches: their latency (H rows in sequence for a sweep, down then
    // up; three passes along 1242-pixel rows for a pair) is 1.0 / 2.0 ms (4 / 8 paths) whatever the frame count,
    // while the line kernels scale with it.  Measured at 1242x375x128 (ms per batch, line vs fused):
    // 8 paths 8 frames 1.96 / 2.20, 12 frames 2.89 / 2.29; 4 paths 8 frames 1.18 / 1.19, 12 frames 1.67 / 1.28.
    const int min_batch = paths == 8 ? par_min_batch(W, H, D) : pairs_min_batch(W, H, D);
    const bool want = agg_mode == 2 || agg_mode == 3 || agg_mode == 6 || (agg_mode == 0 && batch >= min_batch);
    // (P1 <= P2: the fused kernels' form of the step clamps path states at P2 first, epi_sweep.hip)
    const bool fusable = nowrap && P1 <= P2;
    // (the Y volumes hold y + P1 per path since round 3 -- step_b, epi_step.h -- so three / two of them must fit a byte with the bias)
    if (fusable && 3 * (P1 + P2) <= 255 && paths == 8 && want) {
        c.kind = AGG_SWEEP;
        // Between the line kernels and the full pipeline: the down and the up sweep side by side (H rows in sequence
        // instead of 2 H) with Y_up written out and a WTA kernel over C, Y_dn, Y_up, Y_h: 3 B per voxel more traffic,
        // half the latency.  Mode 3 forces it; auto takes it while the batch is too small to hide the longer chain.
        c.sweep_par = agg_mode == 3 || agg_mode == 6 || (agg_mode == 0 && batch < par_max_batch(W, H, D));
        // Mode 6, and auto for the larger of the batches that take the parallel sweeps: the two sweeps meet in the middle.  Each
        // writes its Y for its first half of the rows only and crosses the other's half as a final sweep (the other's Y, Y_h,
        // WTA in registers): the traffic of the full pipeline (9.6 B per voxel measured, no WTA kernel over four volumes) on the
        // parallel sweeps' chain of H rows.
        const int mid_min = mid_min_batch(W, H, D);
        c.sweep_mid = agg_mode == 6 || (agg_mode == 0 && c.sweep_par && mid_min > 0 && batch >= mid_min);
    }
    // the shipped 4-path configuration: both axes as pair kernels, the vertical one final
    if (fusable && 2 * (P1 + P2) <= 255 && paths == 4 && want) c.kind = AGG_PAIRS;
    // very large batches (or mode 4 / 5): the band sweeps
    const int band = agg_mode == 0 ? band_choice(batch, cus, paths, W, H, D) : 0;
    if (fusable && band_ok(D, paths, P1, P2, cm) && (agg_mode == 4 || agg_mode == 5 || band != 0)) {
        c.kind = AGG_BAND;
        c.sweep_par = false;
        c.sweep_mid = false;
        // mode 5 / auto between the sequential form's rounds: the bands of a frame as workgroups of their own (chained)
        c.band_chain = agg_mode == 5 || band == 2;
    }
    return c;
}

static const char* pipeline_name(int kind, bool sweep_par, bool band_chain, bool sweep_mid = false) {
    switch (kind) {
        case AGG_PACKED_NOWRAP: return "packed16/nowrap";
        case AGG_PACKED_WRAP: return "packed16/wrap";
        case AGG_SWEEP: return sweep_mid ? "sweep16mid/nowrap" : sweep_par ? "sweep16par/nowrap" : "sweep16/nowrap";
        case AGG_PAIRS: return "pairs16/nowrap";
        case AGG_BAND: return band_chain ? "band16chain/nowrap" : "band16/nowrap";
        default: return "generic";
    }
}

static void select_kernel(fsgm_epi_plan* p) {
    p->packed = agg_packed_lpp(p->D) != 0;
    const int cm = *std::max_element(p->cmax.begin(), p->cmax.end());
    const PipelineChoice c = choose_pipeline(p->W, p->H, p->D, p->batch, p->prm.paths, p->P1, p->P2, cm, p->agg_mode, p->cus);
    p->kernel_kind = c.kind;
    p->sweep_par = c.sweep_par;
    p->sweep_mid = c.sweep_mid;
    p->band_chain = c.band_chain;
}

// Lazily allocated buffer sets of the two fused pipelines.  Everything is created into locals and committed to
// the plan only when the whole set exists, so a failure midway leaves the plan as it was (nothing leaked,
// nothing half-initialised for the next call to trip over).
namespace {
struct LazySet {
    std::vector<void*> bufs;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
    hipError_t err = hipSuccess;
    template <class T> void alloc(T** p, size_t bytes) {
        *p = nullptr;
        if (err != hipSuccess) return;
        void* v = nullptr;
        err = hipMalloc(&v, bytes ? bytes : 1);
        if (err == hipSuccess) { bufs.push_back(v); *p = (T*)v; }
    }
    void stream(hipStream_t* s) {
        *s = nullptr;
        if (err != hipSuccess) return;
        err = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
        if (err == hipSuccess) streams.push_back(*s);
    }
    void event(hipEvent_t* e) {
        *e = nullptr;
        if (err != hipSuccess) return;
        err = hipEventCreateWithFlags(e, hipEventDisableTiming);
        if (err == hipSuccess) events.push_back(*e);
    }
    void rollback() {
        for (void* b : bufs) (void)hipFree(b);
        for (hipStream_t s : streams) (void)hipStreamDestroy(s);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
};
fsgm_status lazy_fail(LazySet& ls, const char* what) {
    const hipError_t e = ls.err;
    ls.rollback();
    return fail(e == hipErrorOutOfMemory ? FSGM_ERR_NOMEM : FSGM_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
}  // namespace

extern "C" {

const char* fsgm_last_error(void) { return last_error_buf(); }

int fsgm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

fsgm_status fsgm_device_arch(int device, char* buf, size_t buflen) {
    FSGM_REQUIRE(buf && buflen > 0, "fsgm_device_arch: null buffer");
    hipDeviceProp_t prop;
    FSGM_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return FSGM_OK;
}

fsgm_epi_params fsgm_epi_params_default(void) {
    fsgm_epi_params p;
    p.paths = 4;         // calc_cost_sgm.cpp:104
    p.subpixel = 1;      // calc_cost_sgm.cpp:560
    p.vz_to_disp = 1;    // calc_cost_sgm.cpp:4
    p.device = 0;
    p.fb_check = 0;      // calc_cost_sgm.cpp:589-590 (commented out)
    return p;
}

void fsgm_epi_plan_destroy(fsgm_epi_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->prm.device);
    void* bufs[] = {p->dI1, p->dI2, p->dCen1, p->dCen2, p->dPd0, p->dNd, p->dOff, p->dVz,
                    p->dCraw, p->dC, p->dL, p->dBestD, p->dMinC, p->dS, p->dD2enc, p->dD2, p->dConf, p->dLh, p->dX, p->dXup, p->dXupAll, p->dStateUp, p->dLx, p->dState, p->dCkpt, p->dCkptV, p->dRec, p->dS0, p->dRflow, p->dFlow, p->dRgb, p->dBits, p->dBandEdge, p->dBandTicket, p->dBandErr};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    for (hipEvent_t e : {p->ev_fork, p->ev_h, p->ev_b, p->ev_c, p->ev_hl[0], p->ev_hl[1], p->ev_hl[2]})
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {p->stream, p->stream_h, p->stream_b, p->stream_c})
        if (st) (void)hipStreamDestroy(st);
    delete p;
}

fsgm_status fsgm_epi_plan_create(fsgm_epi_plan** out, int32_t W, int32_t H, int32_t D, int32_t batch,
                                 const fsgm_epi_params* prm) {
    FSGM_REQUIRE(out, "fsgm_epi_plan_create: null plan pointer");
    *out = nullptr;
    FSGM_REQUIRE(W >= 1 && H >= 1, "fsgm_epi_plan_create: width/height must be >= 1 (got %d x %d)", W, H);
    FSGM_REQUIRE(D >= 1, "fsgm_epi_plan_create: dMax must be >= 1 (got %d)", D);
    FSGM_REQUIRE(batch >= 1, "fsgm_epi_plan_create: batch must be >= 1");
    const fsgm_epi_params pr = prm ? *prm : fsgm_epi_params_default();
    FSGM_REQUIRE(pr.paths == 4 || pr.paths == 8, "fsgm_epi_plan_create: paths must be 4 or 8 (got %d)", pr.paths);
    if (pr.fb_check && D > 511)
        return fail(FSGM_ERR_UNSUPPORTED, "fb_check needs dMax <= 511 (bestD must stay below INVALID_DISPARITY)");
    if (D > FSGM_GENERIC_MAX_D)
        return fail(FSGM_ERR_UNSUPPORTED, "dMax %d exceeds the supported maximum %d", D, FSGM_GENERIC_MAX_D);
    if ((double)W * H * D >= 2147483648.0)
        return fail(FSGM_ERR_UNSUPPORTED, "cost volume %d x %d x %d exceeds 2^31 voxels per frame", W, H, D);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FSGM_ERR_HIP, "no HIP device available (libfsgm_hip has no CPU fallback)");
    FSGM_REQUIRE(pr.device >= 0 && pr.device < ndev, "device %d out of range (have %d)", pr.device, ndev);
    FSGM_HIP(hipSetDevice(pr.device));

    fsgm_epi_plan* p = new fsgm_epi_plan;
    p->W = W; p->H = H; p->D = D; p->batch = batch; p->prm = pr;
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, pr.device) == hipSuccess && n > 0) p->cus = n; }
    p->NP = (size_t)W * H; p->N = p->NP * D;
    p->cmax.assign(batch, 24);           // census 5x5: 24 informative bits
    const size_t B = batch;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** ptr, size_t bytes) { if (e == hipSuccess) e = hipMalloc(ptr, bytes); };
    // images, census codes, the two fp64 coordinate maps and the raw cost volume belong to the cost stage and are allocated
    // on first use (ensure_cost_buffers): an aggregation-only plan holds C, the offsets and its pipeline's volumes only
    alloc((void**)&p->dOff, B * p->NP * 8);
    alloc((void**)&p->dVz, (size_t)D * 8);
    alloc((void**)&p->dC, B * p->N);
    // the per-voxel intermediates of the two aggregation strategies (L_r for the line kernels;
    // L_left/right, X_dn, states, records for the fused sweeps) are allocated on first use
    alloc((void**)&p->dBestD, B * p->NP * 4);
    alloc((void**)&p->dMinC, B * p->NP * 4);
    if (pr.fb_check) {
        alloc((void**)&p->dD2enc, B * p->NP * 4);
        alloc((void**)&p->dD2, B * p->NP * 4);
        alloc((void**)&p->dConf, B * p->NP);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    if (e == hipSuccess) e = hipMemsetAsync(p->dOff, 0, B * p->NP * 8, p->stream);
    if (e != hipSuccess) {
        fsgm_epi_plan_destroy(p);
        return fail(e == hipErrorOutOfMemory ? FSGM_ERR_NOMEM : FSGM_ERR_HIP,
                    "fsgm_epi_plan_create: %s", hipGetErrorString(e));
    }
    // once per device: the fused kernels' packed 3-input max / min must be exact u16 operations (epi_sweep.hip)
    {
        static std::mutex mu;
        static int state[64] = {0};                          // 0 unknown, 1 good, 2 bad
        std::lock_guard<std::mutex> lk(mu);
        int& s = state[pr.device & 63];
        if (s == 0) {
            int r = fused_step_selftest(p->stream);
            if (r == 0) r = costbox_selftest(p->stream);         // the fused cost kernel's mean as one fp16 multiply on denormal patterns
            if (r < 0) { fsgm_epi_plan_destroy(p); return fail(FSGM_ERR_HIP, "fsgm_epi_plan_create: self-test of the packed step could not run"); }
            s = r == 0 ? 1 : 2;
        }
        if (s == 2) {
            fsgm_epi_plan_destroy(p);
            return fail(FSGM_ERR_UNSUPPORTED, "this build's v_pk_maximum3_f16 / v_pk_minimum3_f16 / v_pk_mul_f16 do not act as exact u16 operations on denormal "
                                              "patterns (toolchain or float-mode change): the fused kernels would be wrong");
        }
    }
    select_kernel(p);
    *out = p;
    return FSGM_OK;
}

// Buffers of the cost stage (census x2 -> raw costs -> box mean) and of everything else that reads the image pair or the
// coordinate maps: created together on first use, committed only when the whole set exists.
static fsgm_status ensure_cost_buffers(fsgm_epi_plan* p) {
    if (p->dCraw) return FSGM_OK;                                // the set's own marker: created last
    const size_t B = p->batch;
    LazySet ls;
    uint8_t *i1, *i2, *craw; uint32_t *c1, *c2; double *pd0, *nd;
    ls.alloc(&i1, B * p->NP);
    ls.alloc(&i2, B * p->NP);
    ls.alloc(&c1, B * p->NP * 4);
    ls.alloc(&c2, B * p->NP * 4);
    ls.alloc(&pd0, B * p->NP * 16);
    ls.alloc(&nd, B * p->NP * 16);
    ls.alloc(&craw, B * p->N);
    if (ls.err != hipSuccess) return lazy_fail(ls, "cost stage buffers");
    p->dI1 = i1; p->dI2 = i2; p->dCen1 = c1; p->dCen2 = c2; p->dPd0 = pd0; p->dNd = nd; p->dCraw = craw;
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_set_penalties(fsgm_epi_plan* p, int32_t P1, int32_t P2, double vMax) {
    FSGM_REQUIRE(p, "null plan");
    p->P1 = P1; p->P2 = P2;
    if (vMax != p->vMax) p->vz_valid = false;
    p->vMax = vMax;
    select_kernel(p);
    return FSGM_OK;
}

static fsgm_status ensure_vz(fsgm_epi_plan* p) {
    if (p->vz_valid) return FSGM_OK;
    // calc_cost_sgm.cpp:339,360-361 -- depends on d only; same fp64 expressions, host side
    std::vector<double> vz(p->D);
    double vzmax = 0.0;
    const double n = p->D + 1;
    for (int d = 0; d < p->D; d++) {
        const double vzRatio = 1.0 * d / n * p->vMax;
        vz[d] = vzRatio / (1 - vzRatio);
        vzmax = std::isfinite(vz[d]) ? std::max(vzmax, std::fabs(vz[d])) : INFINITY;
    }
    FSGM_HIP(hipMemcpyAsync(p->dVz, vz.data(), (size_t)p->D * 8, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));   // vz is a stack-lifetime host buffer
    p->vzmax = vzmax;
    p->vz_valid = true;
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_upload(fsgm_epi_plan* p, int32_t f, const uint8_t* I1, const uint8_t* I2,
                                 const double* pd0, const double* nd, const double* off) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_REQUIRE(I1 && I2 && pd0 && nd && off, "fsgm_epi_plan_upload: null input");
    FSGM_HIP(hipSetDevice(p->prm.device));
    { fsgm_status cs = ensure_cost_buffers(p); if (cs != FSGM_OK) return cs; }
    const size_t NP = p->NP;
    StreamGuard guard(p->stream);
    FSGM_HIP(hipMemcpyAsync(p->dI1 + f * NP, I1, NP, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(p->dI2 + f * NP, I2, NP, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(p->dPd0 + f * 2 * NP, pd0, NP * 16, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(p->dNd + f * 2 * NP, nd, NP * 16, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(p->dOff + f * NP, off, NP * 8, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));   // pageable host memory: keep the caller's buffers free to reuse
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_upload_cost(fsgm_epi_plan* p, int32_t f, const uint8_t* C) {
    FSGM_REQUIRE(p && C, "fsgm_epi_plan_upload_cost: null argument");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->prm.device));
    int cm = 0;
    for (size_t i = 0; i < p->N; i++) cm = C[i] > cm ? C[i] : cm;
    p->cmax[f] = cm;
    select_kernel(p);
    FSGM_HIP(hipMemcpyAsync(p->dC + f * p->N, C, p->N, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    return FSGM_OK;
}

// frame dst <- frame src of the resident cost volumes, columns rotated by roll_cols (dst[y][(x + roll) % W] = src[y][x]):
// how bench.py fills a large batch with distinct volumes without pushing each through PCIe
fsgm_status fsgm_epi_plan_copy_cost(fsgm_epi_plan* p, int32_t dst, int32_t src, int32_t roll_cols) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(dst >= 0 && dst < p->batch && src >= 0 && src < p->batch && dst != src, "fsgm_epi_plan_copy_cost: bad frame pair %d <- %d", dst, src);
    FSGM_HIP(hipSetDevice(p->prm.device));
    const size_t pitch = (size_t)p->W * p->D;
    const int s = ((roll_cols % p->W) + p->W) % p->W;
    uint8_t* d0 = p->dC + (size_t)dst * p->N;
    const uint8_t* s0 = p->dC + (size_t)src * p->N;
    if (s == 0) {
        FSGM_HIP(hipMemcpyAsync(d0, s0, p->N, hipMemcpyDeviceToDevice, p->stream));
    } else {
        FSGM_HIP(hipMemcpy2DAsync(d0 + (size_t)s * p->D, pitch, s0, pitch, (size_t)(p->W - s) * p->D, p->H, hipMemcpyDeviceToDevice, p->stream));
        FSGM_HIP(hipMemcpy2DAsync(d0, pitch, s0 + (size_t)(p->W - s) * p->D, pitch, (size_t)s * p->D, p->H, hipMemcpyDeviceToDevice, p->stream));
    }
    p->cmax[dst] = p->cmax[src];
    select_kernel(p);
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_upload_offset(fsgm_epi_plan* p, int32_t f, const double* off) {
    FSGM_REQUIRE(p && off, "fsgm_epi_plan_upload_offset: null argument");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->prm.device));
    FSGM_HIP(hipMemcpyAsync(p->dOff + f * p->NP, off, p->NP * 8, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    return FSGM_OK;
}

// 4-path pipeline: the along-x pair as pairx_* kernels (8 costs a lane).  Small batches wait for that pair's serial chain,
// which the finer split shortens (2 / 9 frames at 1242x375x128: 0.96 -> 0.82 / 1.24 -> 1.12 ms); from 16 frames the
// pipeline is bound by its HBM traffic (7.5 B per voxel at ~5 TB/s) and the coarser kernels' fewer instructions win
// (40 frames: 3.47 against 3.68 ms).  FSGM_PAIR_XFINE=0 / 1: never / always (A/B switch).
static int pairs_x_fine(const fsgm_epi_plan* p) {
    static const int env = [] { const char* e = getenv("FSGM_PAIR_XFINE"); return (e && *e) ? atoi(e) : -1; }();
    if (!pair_x_fine_ok(p->D) || env == 0) return 0;
    return env == 1 || p->batch < 16 ? 1 : 0;
}

// Parallel sweeps (8 paths, 5..17 frames): what the batch waits for are serial chains -- the along-x pair's 3 x W steps and
// the sweeps' H / 16 launches.  FSGM_EPI_PAR_FINE / FSGM_EPI_PAR_TALL = 0 / 1 force the two shortenings off / on (A/B).
static int par_pair_fine(const fsgm_epi_plan* p) {
    static const int env = env_int("FSGM_EPI_PAR_FINE", -1);
    if (!pair_x_fine_ok(p->D) || env == 0) return 0;
    return env == 1 || p->batch <= 10 ? 1 : 0;       // 8 frames 1.46 -> 1.32 ms with both; from 12 frames neither pays
}
static int par_tall(const fsgm_epi_plan* p) {
    static const int env = env_int("FSGM_EPI_PAR_TALL", -1);
    if (env >= 0) return env != 0;
    return p->batch <= 10 ? 1 : 0;
}

// the final halves of the sweeps that meet in the middle as 8-wave workgroups too (FSGM_EPI_MID_TALL: A/B switch)
static int mid_tall(const fsgm_epi_plan* p) {
    static const int env = env_int("FSGM_EPI_MID_TALL", -1);
    if (env >= 0) return env != 0;
    return par_tall(p);
}

// Parallel sweeps (not the form that meets in the middle, whose final sweeps read Y_h): the along-x pair as two line-kernel slots.
// FSGM_EPI_PAR_XLINES: 0 never, 1 whenever possible; default: up to 5 frames -- 4 frames 0.871 -> 0.832 ms, 6 frames 0.970 -> 1.050:
// with more frames the two volumes' bytes and the lines' instructions cost more than the pair's longer chain
// (profiles/r04_par_xlines.txt).
static int par_x_lines(const fsgm_epi_plan* p) {
    static const int env = env_int("FSGM_EPI_PAR_XLINES", -1);
    if (!p->sweep_par || p->sweep_mid || env == 0) return 0;
    return env == 1 || p->batch <= 5 ? 1 : 0;
}

// (the pipelines share some buffers -- records, S[0] words, Y volumes, the pair's stream -- and a plan may be switched from
// one to another: each set creates only what is still missing)
static fsgm_status ensure_pairs_buffers(fsgm_epi_plan* p) {
    if (p->dCkptV && p->dLh && p->dCkpt && p->dRec && p->dS0 && p->stream_h) return FSGM_OK;
    const size_t B = p->batch;
    LazySet ls;
    uint8_t *lh = p->dLh, *ck = p->dCkpt, *ckv = p->dCkptV; uint2* rec = p->dRec; uint16_t* s0 = p->dS0;
    hipStream_t sh = p->stream_h; hipEvent_t ef = p->ev_fork, eh = p->ev_h;
    if (!lh) ls.alloc(&lh, B * p->N);
    if (!ck) ls.alloc(&ck, B * pair_ckpt_bytes(p->W, p->H, p->D, 0));
    if (!ckv) ls.alloc(&ckv, B * pair_ckpt_bytes(p->W, p->H, p->D, 1));
    if (!rec) ls.alloc(&rec, B * p->NP * sizeof(uint2));
    if (!s0) ls.alloc(&s0, B * p->NP * sizeof(uint16_t));
    if (!sh) ls.stream(&sh);
    if (!ef) ls.event(&ef);
    if (!eh) ls.event(&eh);
    if (ls.err != hipSuccess) return lazy_fail(ls, "pair pipeline buffers");
    p->dLh = lh; p->dCkpt = ck; p->dCkptV = ckv; p->dRec = rec; p->dS0 = s0;
    p->stream_h = sh; p->ev_fork = ef; p->ev_h = eh;
    return FSGM_OK;
}

static fsgm_status ensure_sweep_buffers(fsgm_epi_plan* p) {
    if (p->dState) return FSGM_OK;                               // the set's own marker: created last
    const size_t B = p->batch;
    const size_t state_stride = sweep_state_bytes(p->W, p->D);
    LazySet ls;
    uint8_t *lh = p->dLh, *ck = p->dCkpt, *state = nullptr, *x = p->dX; uint2* rec = p->dRec; uint16_t* s0 = p->dS0;
    hipStream_t sh = p->stream_h, sb = p->stream_b, sc = p->stream_c;
    hipEvent_t ef = p->ev_fork, eh = p->ev_h, eb = p->ev_b, ec = p->ev_c, ehl[3] = {p->ev_hl[0], p->ev_hl[1], p->ev_hl[2]};
    if (!lh) ls.alloc(&lh, B * p->N);
    if (!ck) ls.alloc(&ck, B * pair_ckpt_bytes(p->W, p->H, p->D, 0));
    if (!rec) ls.alloc(&rec, B * p->NP * sizeof(uint2));
    if (!s0) ls.alloc(&s0, B * p->NP * sizeof(uint16_t));
    if (!x) ls.alloc(&x, B * p->N);
    if (!sh) ls.stream(&sh);
    if (!sb) ls.stream(&sb);
    if (!sc) ls.stream(&sc);
    if (!ef) ls.event(&ef);
    if (!eh) ls.event(&eh);
    if (!eb) ls.event(&eb);
    if (!ec) ls.event(&ec);
    for (int l = 0; l < 3; l++) if (!ehl[l]) ls.event(&ehl[l]);
    ls.alloc(&state, 2 * B * state_stride);
    if (ls.err != hipSuccess) return lazy_fail(ls, "sweep pipeline buffers");
    p->state_stride = state_stride;
    p->dLh = lh; p->dCkpt = ck; p->dState = state; p->dRec = rec; p->dS0 = s0; p->dX = x;
    p->stream_h = sh; p->stream_b = sb; p->stream_c = sc;
    p->ev_fork = ef; p->ev_h = eh; p->ev_b = eb; p->ev_c = ec;
    for (int l = 0; l < 3; l++) p->ev_hl[l] = ehl[l];
    { const char* e = getenv("FSGM_EPI_PAIRSPLIT"); p->pair_split = !(e && *e && atoi(e) == 0); }
    { const char* e = getenv("FSGM_EPI_LANES"); const int v = (e && *e) ? atoi(e) : 2; p->lanes = v < 1 ? 1 : (v > 3 ? 3 : v); }
    return FSGM_OK;
}

static fsgm_status ensure_band_buffers(fsgm_epi_plan* p) {
    const int R = band_rows(p->D);
    const size_t nbands = R ? (size_t)(p->H + R - 1) / R : 1;
    const size_t maps = p->band_chain ? (nbands > 1 ? nbands - 1 : 1) : 1;
    if (p->dBandEdge && p->band_edge_maps >= maps && p->dX && p->dRec && p->dS0 && (!p->band_chain || p->dBandTicket)) return FSGM_OK;
    const size_t B = p->batch;
    LazySet ls;
    uint8_t* x = p->dX; uint2* rec = p->dRec; uint4* edge = p->dBandEdge; uint16_t* s0 = p->dS0; uint32_t *bits = p->dBits, *ticket = p->dBandTicket, *err = p->dBandErr;
    const bool new_edge = !edge || p->band_edge_maps < maps;
    const size_t edge_bytes = B * maps * band_edge_uint4s(p->W, p->D, 8) * sizeof(uint4);
    if (!x) ls.alloc(&x, B * p->N);
    if (!rec) ls.alloc(&rec, B * p->NP * sizeof(uint2));
    if (!s0) ls.alloc(&s0, B * p->NP * sizeof(uint16_t));
    if (new_edge) ls.alloc(&edge, edge_bytes);
    if (!bits && p->prm.paths == 8) ls.alloc(&bits, B * band_bits_u32s(p->W, p->H, p->D) * sizeof(uint32_t));
    if (p->band_chain && !ticket) { ls.alloc(&ticket, sizeof(uint32_t)); ls.alloc(&err, sizeof(uint32_t)); }
    // chained form: hand-off dwords carry a launch tag in their bytes' top bits; all ones = "older than any launch"
    if (ls.err == hipSuccess && new_edge && p->band_chain) ls.err = hipMemsetAsync(edge, 0xFF, edge_bytes, p->stream);
    if (ls.err == hipSuccess && p->band_chain && !p->dBandErr) ls.err = hipMemsetAsync(err, 0, sizeof(uint32_t), p->stream);
    if (ls.err != hipSuccess) return lazy_fail(ls, "band pipeline buffers");
    if (new_edge && p->dBandEdge) { (void)hipStreamSynchronize(p->stream); (void)hipFree(p->dBandEdge); }
    if (new_edge) p->band_salt = 0;
    p->dX = x; p->dRec = rec; p->dS0 = s0; p->dBandEdge = edge; p->band_edge_maps = new_edge ? maps : p->band_edge_maps; p->dBits = bits;
    p->dBandTicket = ticket; p->dBandErr = err;
    return FSGM_OK;
}

static fsgm_status ensure_par_buffers(fsgm_epi_plan* p) {
    if (p->dXupAll && (p->dLx || !par_x_lines(p))) return FSGM_OK;
    LazySet ls;
    uint8_t *xu = p->dXupAll, *su = p->dStateUp, *lx = p->dLx;
    if (!xu) ls.alloc(&xu, (size_t)p->batch * p->N);
    if (!su) ls.alloc(&su, 2 * (size_t)p->batch * sweep_state_bytes(p->W, p->D));
    if (!lx && par_x_lines(p)) ls.alloc(&lx, 2 * (size_t)p->batch * p->N);
    if (ls.err != hipSuccess) return lazy_fail(ls, "parallel sweep buffers");
    p->dXupAll = xu; p->dStateUp = su; p->dLx = lx;
    return FSGM_OK;
}

// What a run of `stages` needs before anything is queued: the cost stage rewrites C with census costs
// (values <= 24), so the bound of the cost values -- and with it the kernel selection -- is settled first;
// then the buffer set of the selected pipeline.
static fsgm_status prepare(fsgm_epi_plan* p, int stages) {
    if ((stages & FSGM_STAGE_COST) || p->prm.fb_check) {
        fsgm_status cs = ensure_cost_buffers(p);
        if (cs != FSGM_OK) return cs;
    }
    if (stages & FSGM_STAGE_COST) {
        bool changed = false;
        for (int& c : p->cmax) { if (c != 24) changed = true; c = 24; }
        if (changed) select_kernel(p);
    }
    if (stages & (FSGM_STAGE_AGGREGATE | FSGM_STAGE_WTA)) {
        if (p->kernel_kind == AGG_SWEEP) {
            fsgm_status st = ensure_sweep_buffers(p);
            if (st == FSGM_OK && p->sweep_par) st = ensure_par_buffers(p);
            return st;
        }
        if (p->kernel_kind == AGG_PAIRS) return ensure_pairs_buffers(p);
        if (p->kernel_kind == AGG_BAND) return ensure_band_buffers(p);
    }
    return FSGM_OK;
}

// Where the sweeps that meet in the middle meet: the row count of the down sweep's first half, a whole number of its launches
// (the up sweep's first half, H - hm rows, ends with a short launch).
static int sweep_mid_row(int H, int D) {
    const int t = 2 * sweep_rows_per_launch(D);                  // rows per launch of the 8-wave form
    const int hm = ((H + 1) / 2 + t - 1) / t * t;
    return std::min(hm, H);
}

// census x2 + cost fill + box of frames [f0, f0 + nf) on the plan's stream (ensure_vz done by the caller)
static void enqueue_cost(fsgm_epi_plan* p, int f0, int nf) {
    const size_t NP = p->NP, o = (size_t)f0;
    launch_census(p->stream, p->dI1 + o * NP, p->dCen1 + o * NP, p->W, p->H, nf);
    launch_census(p->stream, p->dI2 + o * NP, p->dCen2 + o * NP, p->W, p->H, nf);
    EpiCostArgs a;
    a.cen1 = p->dCen1 + o * NP; a.cen2 = p->dCen2 + o * NP; a.pd0 = p->dPd0 + o * 2 * NP; a.nd = p->dNd + o * 2 * NP;
    a.off = p->dOff + o * NP; a.vz = p->dVz; a.vzmax = p->vzmax; a.Craw = p->dCraw + o * p->N; a.W = p->W; a.H = p->H; a.D = p->D;
    launch_epi_cost(p->stream, a, p->dC + o * p->N, nf);
}

static fsgm_status enqueue(fsgm_epi_plan* p, int stages) {
    {
        fsgm_status st = prepare(p, stages);
        if (st != FSGM_OK) return st;
    }
    if (stages & FSGM_STAGE_COST) {
        fsgm_status st = ensure_vz(p);
        if (st != FSGM_OK) return st;
        enqueue_cost(p, 0, p->batch);
    }
    if ((stages & FSGM_STAGE_AGGREGATE) && p->kernel_kind == AGG_SWEEP && p->sweep_par) {
        // three independent chains: the horizontal pair (stream_h), the down sweep (here), the up sweep (stream_b)
        FSGM_HIP(hipEventRecord(p->ev_fork, p->stream));
        FSGM_HIP(hipStreamWaitEvent(p->stream_h, p->ev_fork, 0));
        FSGM_HIP(hipStreamWaitEvent(p->stream_b, p->ev_fork, 0));
        const size_t ckb = pair_ckpt_bytes(p->W, p->H, p->D, 0);
        PairArgs h{};
        h.C = p->dC; h.c_frame_stride = p->N; h.X = p->dLh; h.x_frame_stride = p->N;
        h.ckpt = p->dCkpt; h.ckpt_frame_stride = ckb;
        h.W = p->W; h.H = p->H; h.D = p->D; h.P1 = p->P1; h.P2 = p->P2;
        // small batches wait for the pair's serial chain (3 x W steps): 8 costs a lane shorten it (Y_h then in natural d order)
        if (par_x_lines(p)) {
            // few frames: the two along-x paths as line kernels (hand-written step, chains of W steps instead of the pair's 3 W;
            // L_fwd and L_bwd written out: 2 B per voxel more than Y_h, added up by the WTA kernel)
            AggArgs ax;
            ax.C = p->dC; ax.L = p->dLx; ax.c_frame_stride = p->N; ax.l_frame_stride = 2 * p->N; ax.l_dir_stride = p->N;
            ax.W = p->W; ax.H = p->H; ax.D = p->D; ax.P1 = p->P1; ax.P2 = p->P2;
            launch_aggregate(p->stream_h, ax, 2, p->batch, AGG_PACKED_NOWRAP);
        }
        else if (par_pair_fine(p)) launch_pair_x_fine(p->stream_h, h, p->batch);
        else                  launch_pair(p->stream_h, h, p->batch, 0, false);
        FSGM_HIP(hipEventRecord(p->ev_h, p->stream_h));
        SweepArgs w{};
        w.C = p->dC; w.c_frame_stride = p->N;
        w.X = p->dX; w.x_frame_stride = p->N;
        w.Lh = nullptr; w.lh_frame_stride = 0; w.rec = nullptr; w.s0 = nullptr;
        w.state_in = w.state_out = p->dState; w.state_frame_stride = p->state_stride;
        w.W = p->W; w.H = p->H; w.D = p->D; w.P1 = p->P1; w.P2 = p->P2; w.y0 = 0; w.rows = 0;
        const int tall = par_tall(p);                                 // 8-wave workgroups: half the launches of a sweep
        if (p->sweep_mid) {
            // the sweeps meet in the middle: rows [0, hm) of the frame belong to the down sweep's first half, rows [hm, H) to the
            // up sweep's (its rows [0, H - hm) of the mirrored frame); then each crosses the other's half as a final sweep
            const int hm = sweep_mid_row(p->H, p->D);
            int par_dn = 0, par_up = 0;
            SweepArgs up = w;
            up.state_in = up.state_out = p->dStateUp;
            w.X = p->dX; up.X = p->dXupAll;
            launch_sweep_rows(p->stream, w, p->batch, 0, tall, 0, hm, &par_dn);                  // -> Y_dn of rows [0, hm)
            FSGM_HIP(hipEventRecord(p->ev_c, p->stream));
            launch_sweep_rows(p->stream_b, up, p->batch, 1, tall, 0, p->H - hm, &par_up);          // -> Y_up of rows [hm, H)
            FSGM_HIP(hipEventRecord(p->ev_b, p->stream_b));
            w.Lh = up.Lh = p->dLh; w.lh_frame_stride = up.lh_frame_stride = p->N; w.lh_natural = up.lh_natural = par_pair_fine(p);
            w.rec = up.rec = p->dRec; w.s0 = up.s0 = p->dS0;
            w.X = p->dXupAll; up.X = p->dX;                                                      // what a final sweep reads: the other's Y
            FSGM_HIP(hipStreamWaitEvent(p->stream, p->ev_h, 0));
            FSGM_HIP(hipStreamWaitEvent(p->stream, p->ev_b, 0));
            launch_sweep_rows(p->stream, w, p->batch, 3, mid_tall(p), hm, p->H, &par_dn);                  // rows [hm, H): + Y_up + Y_h, WTA
            FSGM_HIP(hipStreamWaitEvent(p->stream_b, p->ev_h, 0));
            FSGM_HIP(hipStreamWaitEvent(p->stream_b, p->ev_c, 0));
            launch_sweep_rows(p->stream_b, up, p->batch, 2, mid_tall(p), p->H - hm, p->H, &par_up);          // rows [0, hm): + Y_dn + Y_h, WTA
            FSGM_HIP(hipEventRecord(p->ev_hl[0], p->stream_b));
            FSGM_HIP(hipStreamWaitEvent(p->stream, p->ev_hl[0], 0));
        } else {
            launch_sweep(p->stream, w, p->batch, 0, tall);                // pass-0 paths from above -> Y_dn
            w.X = p->dXupAll; w.state_in = w.state_out = p->dStateUp;
            launch_sweep(p->stream_b, w, p->batch, 1, tall);              // pass-1 paths -> Y_up
            FSGM_HIP(hipEventRecord(p->ev_b, p->stream_b));
            FSGM_HIP(hipStreamWaitEvent(p->stream, p->ev_h, 0));
            FSGM_HIP(hipStreamWaitEvent(p->stream, p->ev_b, 0));
        }
    } else if ((stages & FSGM_STAGE_AGGREGATE) && p->kernel_kind == AGG_SWEEP) {
        // One sweep launch (strips x frames workgroups) cannot fill 256 CUs, so the work is forked:
        // the horizontal pair runs on stream_h, and the frames split into two lanes, each sweeping
        // down and then up (the final up sweep needs its lane's X_dn and the horizontal pair).
        const int NLN = std::min(p->lanes, p->batch);
        hipStream_t lane_stream[3] = {p->stream, p->stream_b, p->stream_c};
        hipEvent_t lane_done[3] = {nullptr, p->ev_b, p->ev_c};
        FSGM_HIP(hipEventRecord(p->ev_fork, p->stream));
        FSGM_HIP(hipStreamWaitEvent(p->stream_h, p->ev_fork, 0));
        for (int l = 1; l < NLN; l++) FSGM_HIP(hipStreamWaitEvent(lane_stream[l], p->ev_fork, 0));
        // the two horizontal paths as one sum Y_h, lane by lane: a lane's final sweep waits for its own frames only
        for (int lane = 0, f0 = 0; lane < NLN; lane++) {
            const int nf = p->pair_split ? p->batch / NLN + (lane < p->batch % NLN ? 1 : 0) : p->batch;
            const size_t ckb = pair_ckpt_bytes(p->W, p->H, p->D, 0);
            PairArgs h{};
            h.C = p->dC + (size_t)f0 * p->N; h.c_frame_stride = p->N; h.X = p->dLh + (size_t)f0 * p->N; h.x_frame_stride = p->N;
            h.ckpt = p->dCkpt + (size_t)f0 * ckb; h.ckpt_frame_stride = ckb;
            h.W = p->W; h.H = p->H; h.D = p->D; h.P1 = p->P1; h.P2 = p->P2;
            launch_pair(p->stream_h, h, nf, 0, false);
            FSGM_HIP(hipEventRecord(p->ev_hl[lane], p->stream_h));
            f0 += nf;
            if (!p->pair_split) { for (int l = 1; l < NLN; l++) FSGM_HIP(hipEventRecord(p->ev_hl[l], p->stream_h)); break; }
        }
        for (int lane = 0, f0 = 0; lane < NLN; lane++) {
            const int nf = p->batch / NLN + (lane < p->batch % NLN ? 1 : 0);
            hipStream_t st = lane_stream[lane];
            SweepArgs w{};
            w.C = p->dC + (size_t)f0 * p->N; w.c_frame_stride = p->N;
            w.X = p->dX + (size_t)f0 * p->N; w.x_frame_stride = p->N;
            w.Lh = p->dLh + (size_t)f0 * p->N; w.lh_frame_stride = p->N;
            w.rec = p->dRec + (size_t)f0 * p->NP; w.s0 = p->dS0 + (size_t)f0 * p->NP;
            w.state_in = w.state_out = p->dState + (size_t)2 * f0 * p->state_stride;
            w.state_frame_stride = p->state_stride;
            w.W = p->W; w.H = p->H; w.D = p->D; w.P1 = p->P1; w.P2 = p->P2; w.y0 = 0; w.rows = 0;
            launch_sweep(st, w, nf, 0);                              // pass-0 paths from above -> Y_dn
            FSGM_HIP(hipStreamWaitEvent(st, p->ev_hl[lane], 0));
            launch_sweep(st, w, nf, 2);                              // pass-1 paths + everything else + WTA
            if (lane) {
                FSGM_HIP(hipEventRecord(lane_done[lane], st));
                FSGM_HIP(hipStreamWaitEvent(p->stream, lane_done[lane], 0));
            }
            f0 += nf;
        }
    } else if ((stages & FSGM_STAGE_AGGREGATE) && p->kernel_kind == AGG_BAND) {
        // all four paths of a raster pass in one sweep, one workgroup per frame: first pass -> Y (+ 9th bits), second pass + WTA
        BandArgs b{};
        b.C = p->dC; b.c_frame_stride = p->N;
        b.Y = p->dX; b.y_frame_stride = p->N;
        b.Yb = p->dBits; b.yb_frame_stride = band_bits_u32s(p->W, p->H, p->D);
        b.edge = p->dBandEdge; b.edge_frame_stride = p->band_edge_maps * band_edge_uint4s(p->W, p->D, 8);
        b.rec = p->dRec; b.s0 = p->dS0; b.Sdbg = nullptr;
        b.W = p->W; b.H = p->H; b.D = p->D; b.P1 = p->P1; b.P2 = p->P2;
        b.chain = p->band_chain ? 1 : 0;
        b.frames = p->batch; b.nbands = (p->H + band_rows(p->D) - 1) / band_rows(p->D);
        b.group = p->batch;                                  // frames whose bands are dealt band-major: all (groups of 16-64 measured: no gain)
        b.ticket = p->dBandTicket; b.err = p->dBandErr;
        if (b.chain && p->band_edge_untagged) {              // words of the sequential form / the S tap could pass for tag 0: all ones is never a tag
            FSGM_HIP(hipMemsetAsync(p->dBandEdge, 0xFF, (size_t)p->batch * b.edge_frame_stride * sizeof(uint4), p->stream));
            p->band_edge_untagged = false;
        }
        if (!b.chain) p->band_edge_untagged = true;
        for (int mode = 0; mode <= 2; mode += 2) {
            if (b.chain) {                                   // the work counter restarts on the stream ahead of every launch; a fresh hand-off tag
                const uint32_t t = p->band_salt++ % 15u;     // 0..14: the all-ones pattern of the initial fill is never a valid tag
                b.tag = ((t & 1u) << 7) | ((t & 2u) << 14) | ((t & 4u) << 21) | ((t & 8u) << 28);
                FSGM_HIP(hipMemsetAsync(b.ticket, 0, sizeof(uint32_t), p->stream));
            }
            launch_band(p->stream, b, p->batch, p->prm.paths, mode);
        }
    } else if ((stages & FSGM_STAGE_AGGREGATE) && p->kernel_kind == AGG_PAIRS) {
        // 4 paths: the horizontal pair -> X_h on stream_h while the vertical pair's checkpoint pass runs
        // here; then the vertical sum pass adds X_h + 4*C and does the WTA (7.5 B per voxel, S never in HBM)
        FSGM_HIP(hipEventRecord(p->ev_fork, p->stream));
        FSGM_HIP(hipStreamWaitEvent(p->stream_h, p->ev_fork, 0));
        PairArgs h{};
        h.C = p->dC; h.c_frame_stride = p->N; h.X = p->dLh; h.x_frame_stride = p->N;
        h.ckpt = p->dCkpt; h.ckpt_frame_stride = pair_ckpt_bytes(p->W, p->H, p->D, 0);
        h.W = p->W; h.H = p->H; h.D = p->D; h.P1 = p->P1; h.P2 = p->P2;
        const int fine = pairs_x_fine(p);                       // the along-x pair with 8 costs a lane: its chain is what this pipeline waits for
        if (fine) launch_pair_x_fine(p->stream_h, h, p->batch);
        else      launch_pair(p->stream_h, h, p->batch, 0, false);
        FSGM_HIP(hipEventRecord(p->ev_h, p->stream_h));
        PairArgs v = h;
        v.xo_natural = fine;
        v.X = nullptr; v.x_frame_stride = 0;
        v.ckpt = p->dCkptV; v.ckpt_frame_stride = pair_ckpt_bytes(p->W, p->H, p->D, 1);
        v.Xother = p->dLh; v.xo_frame_stride = p->N; v.rec = p->dRec; v.s0 = p->dS0; v.nC = 4;
        launch_pair(p->stream, v, p->batch, 1, true, 1);
        FSGM_HIP(hipStreamWaitEvent(p->stream, p->ev_h, 0));
        launch_pair(p->stream, v, p->batch, 1, true, 2);
    } else if (stages & FSGM_STAGE_AGGREGATE) {
        if (!p->dL) FSGM_HIP(hipMalloc((void**)&p->dL, (size_t)p->batch * p->N * p->prm.paths));
        AggArgs a;
        a.C = p->dC; a.L = p->dL;
        a.c_frame_stride = p->N; a.l_frame_stride = p->N * p->prm.paths; a.l_dir_stride = p->N;
        a.W = p->W; a.H = p->H; a.D = p->D; a.P1 = p->P1; a.P2 = p->P2;
        launch_aggregate(p->stream, a, p->prm.paths, p->batch, p->kernel_kind);
    }
    if ((stages & FSGM_STAGE_WTA) && p->kernel_kind == AGG_SWEEP && p->sweep_par && !p->sweep_mid) {
        WtaArgs a;                                   // S = 8 (C + P2) - (Y_dn + Y_up + Y_h), argmin, parabola, vz -> disp
        a.L = nullptr; a.l_frame_stride = 0; a.l_dir_stride = 0;
        a.off = p->dOff; a.bestD = p->dBestD; a.minC = p->dMinC; a.vMax = p->vMax;
        a.W = p->W; a.H = p->H; a.D = p->D; a.ndirs = p->prm.paths;
        a.subpixel = p->prm.subpixel; a.vz_to_disp = p->prm.vz_to_disp && !p->prm.fb_check;
        SweepSumArgs q;
        q.C = p->dC; q.Xdn = p->dX; q.Xup = p->dXupAll; q.v_frame_stride = p->N;
        q.Lh = p->dLh; q.lh_frame_stride = p->N; q.lh_natural = par_pair_fine(p);
        q.nC = 8; q.bias = p->P2 + p->P1; q.Sdbg = nullptr;
        q.Lx = nullptr; q.lx_frame_stride = 0;
        if (par_x_lines(p)) { q.Lh = nullptr; q.Lx = p->dLx; q.lx_frame_stride = 2 * p->N; q.nC = 6; }   // S = 6 (C + bias) - (Y_dn + Y_up) + L_fwd + L_bwd
        launch_wta_sweep(p->stream, a, q, p->batch);
    } else if ((stages & FSGM_STAGE_WTA) && (p->kernel_kind == AGG_SWEEP || p->kernel_kind == AGG_PAIRS || p->kernel_kind == AGG_BAND)) {
        WtaArgs a;                                   // the argmin happened inside the final sweep / pair pass; finish the records
        a.L = nullptr; a.l_frame_stride = 0; a.l_dir_stride = 0;
        a.off = p->dOff; a.bestD = p->dBestD; a.minC = p->dMinC; a.vMax = p->vMax;
        a.W = p->W; a.H = p->H; a.D = p->D; a.ndirs = p->prm.paths;
        a.subpixel = p->prm.subpixel; a.vz_to_disp = p->prm.vz_to_disp && !p->prm.fb_check;
        launch_sweep_finish(p->stream, a, p->dRec, p->dS0, p->batch);
    } else if (stages & FSGM_STAGE_WTA) {
        WtaArgs a;
        a.L = p->dL; a.l_frame_stride = p->N * p->prm.paths; a.l_dir_stride = p->N;
        a.off = p->dOff; a.bestD = p->dBestD; a.minC = p->dMinC; a.vMax = p->vMax;
        a.W = p->W; a.H = p->H; a.D = p->D; a.ndirs = p->prm.paths;
        a.subpixel = p->prm.subpixel; a.vz_to_disp = p->prm.vz_to_disp && !p->prm.fb_check;
        launch_wta(p->stream, a, p->batch, p->packed);
    }
    if ((stages & FSGM_STAGE_WTA) && p->prm.fb_check) {
        // the check sees bestD before the vz conversion (order of calc_cost_sgm.cpp:584-594)
        FbArgs b;
        b.D1 = p->dBestD; b.pd0 = p->dPd0; b.nd = p->dNd; b.off = p->dOff;
        b.D2enc = p->dD2enc; b.D2 = p->dD2; b.conf = p->dConf;
        b.vMax = p->vMax; b.W = p->W; b.H = p->H; b.n = p->D + 1; b.thr = 2;      // :483 thr = 2
        launch_fb_check(p->stream, b, p->batch);
        if (p->prm.vz_to_disp) launch_vz_convert(p->stream, p->dBestD, p->dOff, p->W, p->H, p->D, p->vMax, p->batch);
    }
    FSGM_HIP(hipGetLastError());
    return FSGM_OK;
}

// (Replaying the fused-sweep stage -- ~100 launches on three streams -- as one HIP graph was measured on MI355X / ROCm 7.2, 32 frames,
// same box, alternating runs: 9 % SLOWER, 5.23 vs 4.79 ms per step; the runtime does not overlap the three captured branches as well
// as the three streams do.  Removed in round 3.)
static fsgm_status run_stages(fsgm_epi_plan* p, int stages) { return enqueue(p, stages); }

fsgm_status fsgm_epi_plan_run(fsgm_epi_plan* p, int32_t stages) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE((stages & ~FSGM_STAGE_ALL) == 0 && stages != 0, "bad stage mask %d", stages);
    FSGM_HIP(hipSetDevice(p->prm.device));
    return run_stages(p, stages);
}

fsgm_status fsgm_epi_plan_set_agg_mode(fsgm_epi_plan* p, int32_t mode) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(mode >= 0 && mode <= 6, "agg mode must be 0 (auto), 1 (per-direction kernels), 2 (fused sweeps), 3 (parallel sweeps), 4 (band sweeps), 5 (chained band sweeps) or 6 (sweeps meeting in the middle)");
    p->agg_mode = mode;
    select_kernel(p);
    return FSGM_OK;
}

// the chained band sweeps' bounded hand-off waits raise a device flag instead of hanging: surface it after a sync
static fsgm_status check_handoff(fsgm_epi_plan* p) {
    if (p->dBandErr) {                                           // a chained launch happened at some point (the selection may have moved on since)
        uint32_t e = 0;
        FSGM_HIP(hipMemcpy(&e, p->dBandErr, sizeof(e), hipMemcpyDeviceToHost));
        if (e != 0) {
            (void)hipMemset(p->dBandErr, 0, sizeof(e));
            return fail(FSGM_ERR_HIP, "band sweep: a hand-off between the bands of a frame timed out (results of this run are invalid)");
        }
    }
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_sync(fsgm_epi_plan* p) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_HIP(hipSetDevice(p->prm.device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    return check_handoff(p);
}

fsgm_status fsgm_epi_plan_download(fsgm_epi_plan* p, int32_t f, uint32_t* bestD, uint32_t* minC) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->prm.device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    { fsgm_status hs = check_handoff(p); if (hs != FSGM_OK) return hs; }
    if (bestD) FSGM_HIP(hipMemcpy(bestD, p->dBestD + f * p->NP, p->NP * 4, hipMemcpyDeviceToHost));
    if (minC) FSGM_HIP(hipMemcpy(minC, p->dMinC + f * p->NP, p->NP * 4, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_download_fb(fsgm_epi_plan* p, int32_t f, uint8_t* conf, uint32_t* bestD2) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_REQUIRE(p->prm.fb_check, "the plan was created without fb_check");
    FSGM_HIP(hipSetDevice(p->prm.device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    { fsgm_status hs = check_handoff(p); if (hs != FSGM_OK) return hs; }
    if (conf) FSGM_HIP(hipMemcpy(conf, p->dConf + f * p->NP, p->NP, hipMemcpyDeviceToHost));
    if (bestD2) FSGM_HIP(hipMemcpy(bestD2, p->dD2 + f * p->NP, p->NP * 4, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_download_cost(fsgm_epi_plan* p, int32_t f, uint8_t* C) {
    FSGM_REQUIRE(p && C, "fsgm_epi_plan_download_cost: null argument");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->prm.device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    FSGM_HIP(hipMemcpy(C, p->dC + f * p->N, p->N, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_download_census(fsgm_epi_plan* p, int32_t f, uint32_t* cen1, uint32_t* cen2) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->prm.device));
    FSGM_REQUIRE(p->dCen1, "fsgm_epi_plan_download_census: the cost stage has not run on this plan");
    FSGM_HIP(hipStreamSynchronize(p->stream));
    if (cen1) FSGM_HIP(hipMemcpy(cen1, p->dCen1 + f * p->NP, p->NP * 4, hipMemcpyDeviceToHost));
    if (cen2) FSGM_HIP(hipMemcpy(cen2, p->dCen2 + f * p->NP, p->NP * 4, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_download_sum(fsgm_epi_plan* p, int32_t f, uint32_t* S) {
    FSGM_REQUIRE(p && S, "fsgm_epi_plan_download_sum: null argument");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->prm.device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    { fsgm_status hs = check_handoff(p); if (hs != FSGM_OK) return hs; }
    if (p->kernel_kind == AGG_SWEEP) {
        // S never exists in HBM in sweep mode.  Debug tap: materialise X_up of that frame with a
        // non-final up sweep, then let wta_sweep_kernel rebuild S = X_dn + X_up + 6C + L_left + L_right.
        FSGM_HIP(hipStreamSynchronize(p->stream));
        fsgm_status es = ensure_sweep_buffers(p);
        if (es != FSGM_OK) return es;
        if (!p->dS) FSGM_HIP(hipMalloc((void**)&p->dS, p->N * 4));
        if (!p->dXup) FSGM_HIP(hipMalloc((void**)&p->dXup, p->N));
        SweepArgs w{};
        w.C = p->dC + (size_t)f * p->N; w.c_frame_stride = p->N;
        w.X = p->dXup; w.x_frame_stride = p->N;
        w.Lh = nullptr; w.lh_frame_stride = 0; w.rec = nullptr; w.s0 = nullptr;
        w.state_in = w.state_out = p->dState + (size_t)2 * f * p->state_stride;   // idle now: scratch for one frame
        w.state_frame_stride = p->state_stride;
        w.W = p->W; w.H = p->H; w.D = p->D; w.P1 = p->P1; w.P2 = p->P2; w.y0 = 0; w.rows = 0;
        launch_sweep(p->stream, w, 1, 1);
        if (p->sweep_mid) {                                      // the meeting sweeps leave Y_dn for the upper half of the rows only
            w.X = p->dX + (size_t)f * p->N;
            launch_sweep(p->stream, w, 1, 0);
        }
        WtaArgs a;
        a.L = nullptr; a.l_frame_stride = 0; a.l_dir_stride = 0;
        a.off = p->dOff + f * p->NP; a.bestD = p->dBestD + f * p->NP; a.minC = p->dMinC + f * p->NP; a.vMax = p->vMax;
        a.W = p->W; a.H = p->H; a.D = p->D; a.ndirs = p->prm.paths;
        a.subpixel = p->prm.subpixel; a.vz_to_disp = p->prm.vz_to_disp;
        SweepSumArgs q;
        q.C = p->dC + (size_t)f * p->N; q.Xdn = p->dX + (size_t)f * p->N; q.Xup = p->dXup; q.v_frame_stride = p->N;
        q.Lh = p->dLh + (size_t)f * p->N; q.lh_frame_stride = p->N; q.lh_natural = p->sweep_par ? par_pair_fine(p) : 0;
        q.nC = 8; q.bias = p->P2 + p->P1; q.Sdbg = p->dS;
        q.Lx = nullptr; q.lx_frame_stride = 0;
        if (par_x_lines(p)) { q.Lh = nullptr; q.Lx = p->dLx + (size_t)f * 2 * p->N; q.lx_frame_stride = 2 * p->N; q.nC = 6; }
        launch_wta_sweep(p->stream, a, q, 1);
        FSGM_HIP(hipGetLastError());
        FSGM_HIP(hipStreamSynchronize(p->stream));
        FSGM_HIP(hipMemcpy(S, p->dS, p->N * 4, hipMemcpyDeviceToHost));
        return FSGM_OK;
    }
    if (p->kernel_kind == AGG_PAIRS) {
        // Debug tap of the 4-path pair pipeline: materialise the vertical pair's X_v of that frame with a
        // non-final sum pass, then wta_sweep_kernel rebuilds S = X_v + X_h + 4C.
        FSGM_HIP(hipStreamSynchronize(p->stream));
        fsgm_status es = ensure_pairs_buffers(p);
        if (es != FSGM_OK) return es;
        if (!p->dS) FSGM_HIP(hipMalloc((void**)&p->dS, p->N * 4));
        if (!p->dXup) FSGM_HIP(hipMalloc((void**)&p->dXup, p->N));          // one frame of scratch (dX is a whole-batch buffer of other pipelines)
        PairArgs v{};
        v.C = p->dC + (size_t)f * p->N; v.c_frame_stride = p->N; v.X = p->dXup; v.x_frame_stride = p->N;
        v.ckpt = p->dCkptV; v.ckpt_frame_stride = pair_ckpt_bytes(p->W, p->H, p->D, 1);
        v.W = p->W; v.H = p->H; v.D = p->D; v.P1 = p->P1; v.P2 = p->P2;
        launch_pair(p->stream, v, 1, 1, false);
        WtaArgs a;
        a.L = nullptr; a.l_frame_stride = 0; a.l_dir_stride = 0;
        a.off = p->dOff + f * p->NP; a.bestD = p->dBestD + f * p->NP; a.minC = p->dMinC + f * p->NP; a.vMax = p->vMax;
        a.W = p->W; a.H = p->H; a.D = p->D; a.ndirs = p->prm.paths;
        a.subpixel = p->prm.subpixel; a.vz_to_disp = p->prm.vz_to_disp;
        SweepSumArgs q;
        q.C = p->dC + (size_t)f * p->N; q.Xdn = p->dXup; q.Xup = nullptr; q.v_frame_stride = p->N;
        q.Lh = p->dLh + (size_t)f * p->N; q.lh_frame_stride = p->N; q.lh_natural = pairs_x_fine(p);
        q.nC = 4; q.bias = p->P2 + p->P1; q.Sdbg = p->dS;
        q.Lx = nullptr; q.lx_frame_stride = 0;
        launch_wta_sweep(p->stream, a, q, 1);
        FSGM_HIP(hipGetLastError());
        FSGM_HIP(hipStreamSynchronize(p->stream));
        FSGM_HIP(hipMemcpy(S, p->dS, p->N * 4, hipMemcpyDeviceToHost));
        return FSGM_OK;
    }
    if (p->kernel_kind == AGG_BAND) {
        // S never exists in HBM in band mode either.  Debug tap: the second pass of that frame again (its first pass's Y is
        // still in place), with the kernel's natural-order dump of S switched on.
        FSGM_HIP(hipStreamSynchronize(p->stream));
        fsgm_status es = ensure_band_buffers(p);
        if (es != FSGM_OK) return es;
        if (!p->dS) FSGM_HIP(hipMalloc((void**)&p->dS, p->N * 4));
        BandArgs b{};
        const size_t fs = (size_t)f;
        b.C = p->dC + fs * p->N; b.c_frame_stride = p->N;
        b.Y = p->dX + fs * p->N; b.y_frame_stride = p->N;
        b.yb_frame_stride = band_bits_u32s(p->W, p->H, p->D);
        b.Yb = p->dBits ? p->dBits + fs * b.yb_frame_stride : nullptr;
        b.edge_frame_stride = p->band_edge_maps * band_edge_uint4s(p->W, p->D, 8);
        b.edge = p->dBandEdge + fs * b.edge_frame_stride;
        b.rec = p->dRec + fs * p->NP; b.s0 = p->dS0 + fs * p->NP; b.Sdbg = p->dS;
        b.W = p->W; b.H = p->H; b.D = p->D; b.P1 = p->P1; b.P2 = p->P2;
        launch_band(p->stream, b, 1, p->prm.paths, 2);
        p->band_edge_untagged = true;
        FSGM_HIP(hipGetLastError());
        FSGM_HIP(hipStreamSynchronize(p->stream));
        FSGM_HIP(hipMemcpy(S, p->dS, p->N * 4, hipMemcpyDeviceToHost));
        return FSGM_OK;
    }
    if (!p->dS) FSGM_HIP(hipMalloc((void**)&p->dS, p->N * 4));
    launch_sum_paths(p->stream, p->dL + (size_t)f * p->N * p->prm.paths, p->dS, p->N, p->N, p->prm.paths);
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipStreamSynchronize(p->stream));
    FSGM_HIP(hipMemcpy(S, p->dS, p->N * 4, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_epi_plan_time(fsgm_epi_plan* p, int32_t stages, int32_t warmup, int32_t iters, float* ms_avg) {
    FSGM_REQUIRE(p && ms_avg, "fsgm_epi_plan_time: null argument");
    FSGM_REQUIRE(iters >= 1 && warmup >= 0, "fsgm_epi_plan_time: iters must be >= 1");
    FSGM_REQUIRE((stages & ~FSGM_STAGE_ALL) == 0 && stages != 0, "bad stage mask %d", stages);
    FSGM_HIP(hipSetDevice(p->prm.device));
    for (int i = 0; i < warmup; i++) {
        fsgm_status st = run_stages(p, stages);
        if (st != FSGM_OK) return st;
    }
    FSGM_HIP(hipEventRecord(p->ev0, p->stream));
    for (int i = 0; i < iters; i++) {
        fsgm_status st = run_stages(p, stages);
        if (st != FSGM_OK) return st;
    }
    FSGM_HIP(hipEventRecord(p->ev1, p->stream));
    FSGM_HIP(hipEventSynchronize(p->ev1));
    float ms = 0;
    FSGM_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *ms_avg = ms / iters;
    return check_handoff(p);                                     // a timed run that gave up on a hand-off is not a measurement
}

void* fsgm_epi_plan_stream(fsgm_epi_plan* p) { return p ? (void*)p->stream : nullptr; }

const char* fsgm_epi_plan_kernel_name(fsgm_epi_plan* p) {
    if (!p) return "";
    return pipeline_name(p->kernel_kind, p->sweep_par, p->band_chain, p->sweep_mid);
}

const char* fsgm_epi_auto_pipeline(int32_t width, int32_t height, int32_t dMax, int32_t batch, int32_t paths, int32_t P1, int32_t P2,
                                   int32_t cmax, int32_t cus) {
    if (width <= 0 || height <= 0 || dMax <= 0 || batch <= 0 || (paths != 4 && paths != 8) || cus <= 0) return "";
    const PipelineChoice c = choose_pipeline(width, height, dMax, batch, paths, P1, P2, cmax, 0, cus);
    return pipeline_name(c.kind, c.sweep_par, c.band_chain, c.sweep_mid);
}

// The achievable HBM rate of this device, measured the way the aggregation kernels move bytes: a grid-stride
// copy kernel, 16 B per lane per access (launch_copy16, epi_kernels.hip), device memory to device memory, read + written
// bytes counted.  mode 0: that kernel; mode 1: hipMemcpyAsync D2D (the runtime's blit kernel), for comparison.
fsgm_status fsgm_measure_copy_bandwidth2(int32_t device, size_t bytes, int32_t iters, int32_t mode, double* gbps) {
    FSGM_REQUIRE(gbps && bytes >= 4096 && iters > 0, "fsgm_measure_copy_bandwidth: bad argument");
    FSGM_REQUIRE(mode == 0 || mode == 1, "fsgm_measure_copy_bandwidth: mode must be 0 (copy kernel) or 1 (hipMemcpyAsync)");
    FSGM_HIP(hipSetDevice(device));
    bytes &= ~(size_t)4095;
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0, e1;
    FSGM_HIP(hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); return fail(FSGM_ERR_NOMEM, "copy probe: out of memory"); }
    (void)hipMemset(a, 1, bytes);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto once = [&]() {
        if (mode == 0) launch_copy16(0, b, a, bytes);
        else (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    };
    once();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; i++) once();
    (void)hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipGetLastError();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    if (e != hipSuccess) return fail(FSGM_ERR_HIP, "copy probe: %s", hipGetErrorString(e));
    *gbps = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
    return FSGM_OK;
}

fsgm_status fsgm_measure_copy_bandwidth(int32_t device, size_t bytes, int32_t iters, double* gbps) {
    return fsgm_measure_copy_bandwidth2(device, bytes, iters, 0, gbps);
}

// ---------------------------------------------------------------------------------------------
// host-pointer entry points (the MEX boundary).  Plans are cached per shape for the lifetime of
// the process so repeated MEX calls do not re-allocate HBM (SURVEY 8b "ownership").
// ---------------------------------------------------------------------------------------------
static PerDevice<std::vector<fsgm_epi_plan*>> g_epi;            // cached plans per device, under that device's lock

static fsgm_status cached_plan(fsgm_epi_plan** out, int W, int H, int D, int batch, const fsgm_epi_params& pr) {
    std::vector<fsgm_epi_plan*>& g_cache = g_epi.v[pr.device];   // (the caller holds g_epi.mu[pr.device])
    for (fsgm_epi_plan* p : g_cache)
        if (p->W == W && p->H == H && p->D == D && p->batch == batch && p->prm.paths == pr.paths &&
            p->prm.device == pr.device && p->prm.fb_check == pr.fb_check) {
            p->prm = pr;
            *out = p;
            return FSGM_OK;
        }
    fsgm_epi_plan* p = nullptr;
    fsgm_status st = fsgm_epi_plan_create(&p, W, H, D, batch, &pr);
    if (st != FSGM_OK) return st;
    if (g_cache.size() >= 4) {           // bound the HBM held by stale shapes
        fsgm_epi_plan_destroy(g_cache.front());
        g_cache.erase(g_cache.begin());
    }
    g_cache.push_back(p);
    *out = p;
    return FSGM_OK;
}

void fsgm_pyd_shutdown_internal(void);
void fsgm_pyramid_shutdown_internal(void);
void fsgm_post_shutdown_internal(void);
void fsgm_ng_shutdown_internal(void);
void fsgm_ng_pyramid_shutdown_internal(void);

void fsgm_shutdown(void) {
    fsgm_ng_pyramid_shutdown_internal();
    fsgm_ng_shutdown_internal();
    fsgm_post_shutdown_internal();
    fsgm_pyramid_shutdown_internal();
    fsgm_pyd_shutdown_internal();
    for (int d = 0; d < FSGM_MAX_DEVICES; d++) {
        std::lock_guard<std::mutex> lk(g_epi.mu[d]);
        for (fsgm_epi_plan* p : g_epi.v[d]) fsgm_epi_plan_destroy(p);
        g_epi.v[d].clear();
    }
}

fsgm_status fsgm_calc_cost_sgm_batch_host(int32_t n, const fsgm_epi_in* in, const fsgm_epi_out* out,
                                          const fsgm_epi_params* prm) {
    FSGM_REQUIRE(n >= 1 && in && out, "fsgm_calc_cost_sgm: null argument");
    const fsgm_epi_params pr = prm ? *prm : fsgm_epi_params_default();
    for (int i = 0; i < n; i++) {
        FSGM_REQUIRE(in[i].I1 && in[i].I2 && in[i].pixelPosD0 && in[i].normDir && in[i].offset,
                     "fsgm_calc_cost_sgm: frame %d has a null input", i);
        FSGM_REQUIRE(out[i].bestD && out[i].minC, "fsgm_calc_cost_sgm: frame %d has a null output", i);
        FSGM_REQUIRE(in[i].width == in[0].width && in[i].height == in[0].height && in[i].dMax == in[0].dMax &&
                     in[i].P1 == in[0].P1 && in[i].P2 == in[0].P2 && in[i].vMax == in[0].vMax,
                     "fsgm_calc_cost_sgm: frames of one batch must share shape and parameters (frame %d differs)", i);
    }
    FSGM_DEVICE_SLOT(pr.device);
    std::lock_guard<std::mutex> lk(g_epi.mu[pr.device]);
    fsgm_epi_plan* p = nullptr;
    fsgm_status st = cached_plan(&p, in[0].width, in[0].height, in[0].dMax, n, pr);
    if (st != FSGM_OK) return st;
    if ((st = fsgm_epi_plan_set_penalties(p, in[0].P1, in[0].P2, in[0].vMax)) != FSGM_OK) return st;
    FSGM_HIP(hipSetDevice(p->prm.device));
    if ((st = ensure_cost_buffers(p)) != FSGM_OK) return st;
    // One call = one stream-ordered sequence with a single host wait: every frame's inputs go up asynchronously on the
    // plan's stream (hipMemcpyAsync from the caller's pageable memory runs at the pinned rate here, ~50 GB/s, so there is
    // no staging copy: tools/ubench/h2d_rates.hip), the batched kernels follow, the results come down at the end.
    // Measured alternatives that lost (A/B on one box): uploads on a second stream with each frame's cost stage started
    // as the frame arrives -- the per-frame launches cost more than the overlap saves (1.29 vs 1.20 ms for one
    // 1242x375x128 frame, 9.2 vs 7.9 ms for eight); a pinned staging ring (an extra host copy at 25-34 GB/s).
    StreamGuard guard(p->stream);                                // every early exit drains the stream: the copies use caller memory
    const size_t NP = p->NP;
    for (int i = 0; i < n; i++) {
        FSGM_HIP(hipMemcpyAsync(p->dI1 + i * NP, in[i].I1, NP, hipMemcpyHostToDevice, p->stream));
        FSGM_HIP(hipMemcpyAsync(p->dI2 + i * NP, in[i].I2, NP, hipMemcpyHostToDevice, p->stream));
        FSGM_HIP(hipMemcpyAsync(p->dPd0 + (size_t)i * 2 * NP, in[i].pixelPosD0, NP * 16, hipMemcpyHostToDevice, p->stream));
        FSGM_HIP(hipMemcpyAsync(p->dNd + (size_t)i * 2 * NP, in[i].normDir, NP * 16, hipMemcpyHostToDevice, p->stream));
        FSGM_HIP(hipMemcpyAsync(p->dOff + i * NP, in[i].offset, NP * 8, hipMemcpyHostToDevice, p->stream));
    }
    if ((st = run_stages(p, FSGM_STAGE_ALL)) != FSGM_OK) return st;
    bool taps = false;
    for (int i = 0; i < n; i++) {
        FSGM_HIP(hipMemcpyAsync(out[i].bestD, p->dBestD + i * NP, NP * 4, hipMemcpyDeviceToHost, p->stream));
        FSGM_HIP(hipMemcpyAsync(out[i].minC, p->dMinC + i * NP, NP * 4, hipMemcpyDeviceToHost, p->stream));
        if (pr.fb_check && out[i].conf) FSGM_HIP(hipMemcpyAsync(out[i].conf, p->dConf + i * NP, NP, hipMemcpyDeviceToHost, p->stream));
        if (pr.fb_check && out[i].bestD2) FSGM_HIP(hipMemcpyAsync(out[i].bestD2, p->dD2 + i * NP, NP * 4, hipMemcpyDeviceToHost, p->stream));
        if (out[i].C) FSGM_HIP(hipMemcpyAsync(out[i].C, p->dC + (size_t)i * p->N, p->N, hipMemcpyDeviceToHost, p->stream));
        taps = taps || out[i].S;
    }
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    if ((st = check_handoff(p)) != FSGM_OK) return st;
    if (taps)
        for (int i = 0; i < n; i++)
            if (out[i].S && (st = fsgm_epi_plan_download_sum(p, i, out[i].S)) != FSGM_OK) return st;
    return FSGM_OK;
}

fsgm_status fsgm_calc_cost_sgm_host(const fsgm_epi_in* in, const fsgm_epi_out* out, const fsgm_epi_params* prm) {
    return fsgm_calc_cost_sgm_batch_host(1, in, out, prm);
}

// sgm(C, P1, P2): sgm.m's call shape on the MEX's aggregation + WTA (MEX semantics: include/fsgm.h)
fsgm_status fsgm_sgm_host(const uint8_t* C, int32_t W, int32_t H, int32_t D, int32_t P1, int32_t P2, int32_t paths,
                          uint32_t* bestD, uint32_t* minC, uint32_t* S, int32_t device) {
    FSGM_REQUIRE(C && bestD && minC, "fsgm_sgm: null argument");
    fsgm_epi_params pr = fsgm_epi_params_default();
    pr.paths = paths; pr.device = device; pr.vz_to_disp = 0; pr.subpixel = 1;
    FSGM_DEVICE_SLOT(device);
    std::lock_guard<std::mutex> lk(g_epi.mu[device]);
    fsgm_epi_plan* p = nullptr;
    fsgm_status st = cached_plan(&p, W, H, D, 1, pr);
    if (st != FSGM_OK) return st;
    if ((st = fsgm_epi_plan_set_penalties(p, P1, P2, p->vMax)) != FSGM_OK) return st;
    if ((st = fsgm_epi_plan_upload_cost(p, 0, C)) != FSGM_OK) return st;
    if ((st = fsgm_epi_plan_run(p, FSGM_STAGE_AGGREGATE | FSGM_STAGE_WTA)) != FSGM_OK) return st;
    if ((st = fsgm_epi_plan_download(p, 0, bestD, minC)) != FSGM_OK) return st;
    if (S && (st = fsgm_epi_plan_download_sum(p, 0, S)) != FSGM_OK) return st;
    return FSGM_OK;
}

// census() of common.cpp:3-27 alone (the one function of the path that the reference's own sources pin here)
fsgm_status fsgm_census_host(const uint8_t* img, int32_t W, int32_t H, uint32_t* cen, int32_t device) {
    FSGM_REQUIRE(img && cen, "fsgm_census: null argument");
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1 (got %d x %d)", W, H);
    fsgm_epi_params pr = fsgm_epi_params_default();
    pr.device = device;
    FSGM_DEVICE_SLOT(device);
    std::lock_guard<std::mutex> lk(g_epi.mu[device]);
    fsgm_epi_plan* p = nullptr;
    fsgm_status st = cached_plan(&p, W, H, 16, 1, pr);           // any dMax: only the image / census buffers are used
    if (st != FSGM_OK) return st;
    FSGM_HIP(hipSetDevice(device));
    if ((st = ensure_cost_buffers(p)) != FSGM_OK) return st;
    StreamGuard guard(p->stream);
    FSGM_HIP(hipMemcpyAsync(p->dI1, img, p->NP, hipMemcpyHostToDevice, p->stream));
    launch_census(p->stream, p->dI1, p->dCen1, W, H, 1);
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipMemcpyAsync(cen, p->dCen1, p->NP * 4, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

// ---- the dense half of the epipolar driver (dense part only) ----
static EpiGeomArgs geom_args(const fsgm_epi_geometry* g, int W, int H, double* Pd0, double* nd, double* off, double* rflow) {
    EpiGeomArgs a;
    for (int i = 0; i < 9; i++) { a.F[i] = g->F[i]; a.Hm[i] = g->H[i]; }
    a.ex = g->epipole[0]; a.ey = g->epipole[1]; a.direction = g->direction != 0;
    a.Pd0 = Pd0; a.nd = nd; a.off = off; a.rflow = rflow; a.W = W; a.H = H;
    return a;
}

static fsgm_status ensure_driver_buffers(fsgm_epi_plan* p, int channels) {
    { fsgm_status cs = ensure_cost_buffers(p); if (cs != FSGM_OK) return cs; }
    if (!p->dRflow) FSGM_HIP(hipMalloc((void**)&p->dRflow, (size_t)p->batch * p->NP * 16));
    if (!p->dFlow) FSGM_HIP(hipMalloc((void**)&p->dFlow, (size_t)p->batch * p->NP * 24));
    if (channels == 3 && !p->dRgb) FSGM_HIP(hipMalloc((void**)&p->dRgb, p->NP * 3 * 2));
    return FSGM_OK;
}

fsgm_status fsgm_epipolar_maps_host(const fsgm_epi_geometry* g, int32_t W, int32_t H, double* Pd0, double* normDirect,
                                    double* Offset, double* Rflow, int32_t device) {
    FSGM_REQUIRE(g && Pd0 && normDirect && Offset && Rflow, "fsgm_epipolar_maps: null argument");
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1 (got %d x %d)", W, H);
    fsgm_epi_params pr = fsgm_epi_params_default();
    pr.device = device;
    FSGM_DEVICE_SLOT(device);
    std::lock_guard<std::mutex> lk(g_epi.mu[device]);
    fsgm_epi_plan* p = nullptr;
    fsgm_status st = cached_plan(&p, W, H, 16, 1, pr);           // any dMax: only the map buffers are used
    if (st != FSGM_OK) return st;
    FSGM_HIP(hipSetDevice(device));
    if ((st = ensure_driver_buffers(p, 1)) != FSGM_OK) return st;
    StreamGuard guard(p->stream);
    launch_epi_maps(p->stream, geom_args(g, W, H, p->dPd0, p->dNd, p->dOff, p->dRflow));
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipMemcpyAsync(Pd0, p->dPd0, p->NP * 16, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipMemcpyAsync(normDirect, p->dNd, p->NP * 16, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipMemcpyAsync(Offset, p->dOff, p->NP * 8, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipMemcpyAsync(Rflow, p->dRflow, p->NP * 16, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_epipolar_sgm_of_host(const uint8_t* I0, const uint8_t* I1, int32_t W, int32_t H, int32_t channels,
                                      const fsgm_epi_geometry* g, int32_t dMax, double vMax, const fsgm_epi_params* prm,
                                      double* flow, uint32_t* minC) {
    FSGM_REQUIRE(I0 && I1 && g && flow, "fsgm_epipolar_sgm_of: null argument");
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1 (got %d x %d)", W, H);
    FSGM_REQUIRE(channels == 1 || channels == 3, "channels must be 1 (gray) or 3 (RGB planes), got %d", channels);
    fsgm_epi_params pr = prm ? *prm : fsgm_epi_params_default();
    FSGM_REQUIRE(pr.vz_to_disp && !pr.fb_check, "fsgm_epipolar_sgm_of: the flow needs disparities (vz_to_disp = 1, fb_check = 0)");
    FSGM_DEVICE_SLOT(pr.device);
    std::lock_guard<std::mutex> lk(g_epi.mu[pr.device]);
    fsgm_epi_plan* p = nullptr;
    fsgm_status st = cached_plan(&p, W, H, dMax, 1, pr);
    if (st != FSGM_OK) return st;
    FSGM_HIP(hipSetDevice(pr.device));
    if ((st = fsgm_epi_plan_set_penalties(p, 6, 64, vMax)) != FSGM_OK) return st;        // epipolar_sgm_of.m:19
    if ((st = ensure_driver_buffers(p, channels)) != FSGM_OK) return st;
    const size_t NP = p->NP;
    StreamGuard guard(p->stream);
    if (channels == 3) {                                                                 // epipolar_sgm_of.m:35-38
        FSGM_HIP(hipMemcpyAsync(p->dRgb, I0, 3 * NP, hipMemcpyHostToDevice, p->stream));
        FSGM_HIP(hipMemcpyAsync(p->dRgb + 3 * NP, I1, 3 * NP, hipMemcpyHostToDevice, p->stream));
        launch_pyr_gray(p->stream, p->dRgb, p->dI1, W, H);
        launch_pyr_gray(p->stream, p->dRgb + 3 * NP, p->dI2, W, H);
    } else {
        FSGM_HIP(hipMemcpyAsync(p->dI1, I0, NP, hipMemcpyHostToDevice, p->stream));
        FSGM_HIP(hipMemcpyAsync(p->dI2, I1, NP, hipMemcpyHostToDevice, p->stream));
    }
    launch_epi_maps(p->stream, geom_args(g, W, H, p->dPd0, p->dNd, p->dOff, p->dRflow));  // epipolar_sgm_of.m:24
    if ((st = enqueue(p, FSGM_STAGE_ALL)) != FSGM_OK) return st;                          // :45
    launch_epi_flow(p->stream, p->dBestD, p->dNd, p->dRflow, p->dFlow, W, H);             // :46-51
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipMemcpyAsync(flow, p->dFlow, NP * 24, hipMemcpyDeviceToHost, p->stream));
    if (minC) FSGM_HIP(hipMemcpyAsync(minC, p->dMinC, NP * 4, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

}  // extern "C"
