// capi_pyramid.hip -- C ABI for the pyramidal driver (include/fsgm.h): the level loop of
// pyramidal_sgm.m with every level's images, hint maps and flows resident in HBM.  One host call
// uploads the image pair and downloads the flow; nothing crosses PCIe between levels (the
// reference's loop does one MEX call, i.e. one round trip, per level: pyramidal_sgm.m:37-75).
#include "capi_common.h"
#include "pyd_kernels.h"
#include "pyd_plan.h"
#include "pyramid_kernels.h"
#include <mutex>
#include <string.h>
#include <vector>

using namespace fsgm;

struct fsgm_pyramid_plan {
    int W = 0, H = 0, channels = 1, device = 0, batch = 1;
    fsgm_pyramid_params prm{};
    std::vector<int> Ws, Hs;                         // level l (0-based) size
    std::vector<fsgm_pyd_plan*> lv;                  // one calc_pyd_cost_sgm plan per level
    std::vector<uint8_t*> dP0, dP1;                  // colour pyramids [3][h][w] (channels == 3 only)
    std::vector<double*> dFlow;                      // mvPyd{l}: [2][h][w]
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

extern "C" {

fsgm_pyramid_params fsgm_pyramid_params_default(void) {
    fsgm_pyramid_params p;
    p.numPyd = 5;                      // pyramidal_sgm.m:12
    p.P1 = 6; p.P2 = 32;               // :15-16
    p.aggHalfWinSize = 2;              // :17
    p.verSearchHalfWinSize = 5;        // :18
    p.horSearchHalfWinSize = 5;        // :19
    p.enableDiagonal = 1;              // :20
    p.totalPass = 2;                   // :21
    p.adaptiveP2 = 0;                  // :22
    p.device = 0;
    return p;
}

void fsgm_pyramid_plan_destroy(fsgm_pyramid_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (fsgm_pyd_plan* q : p->lv) fsgm_pyd_plan_destroy(q);
    for (uint8_t* b : p->dP0) if (b) (void)hipFree(b);
    for (uint8_t* b : p->dP1) if (b) (void)hipFree(b);
    for (double* b : p->dFlow) if (b) (void)hipFree(b);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

fsgm_status fsgm_pyramid_plan_create(fsgm_pyramid_plan** out, int32_t W, int32_t H, int32_t channels,
                                     const fsgm_pyramid_params* prm) {
    return fsgm_pyramid_plan_create_batch(out, W, H, channels, prm, 1);
}

// `batch` image pairs resident in one plan: every kernel of a level covers all of them (the level loop stays a
// sequence -- a level needs the level above -- but each of its launches has batch times the work)
fsgm_status fsgm_pyramid_plan_create_batch(fsgm_pyramid_plan** out, int32_t W, int32_t H, int32_t channels,
                                           const fsgm_pyramid_params* prm, int32_t batch) {
    FSGM_REQUIRE(out, "fsgm_pyramid_plan_create: null plan pointer");
    *out = nullptr;
    FSGM_REQUIRE(batch >= 1 && batch <= 4096, "batch must be in 1..4096 (got %d)", batch);
    FSGM_REQUIRE(prm, "fsgm_pyramid_plan_create: null parameters");
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1 (got %d x %d)", W, H);
    FSGM_REQUIRE(channels == 1 || channels == 3, "channels must be 1 (gray) or 3 (RGB planes), got %d", channels);
    FSGM_REQUIRE(prm->numPyd >= 1 && prm->numPyd <= 16, "numPyd must be in 1..16 (got %d)", prm->numPyd);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FSGM_ERR_HIP, "no HIP device available (libfsgm_hip has no CPU fallback)");
    FSGM_REQUIRE(prm->device >= 0 && prm->device < ndev, "device %d out of range (have %d)", prm->device, ndev);
    FSGM_HIP(hipSetDevice(prm->device));
    fsgm_pyramid_plan* p = new fsgm_pyramid_plan;
    p->W = W; p->H = H; p->channels = channels; p->device = prm->device; p->prm = *prm; p->batch = batch;
    const int n = prm->numPyd;
    p->Ws.resize(n); p->Hs.resize(n);
    p->Ws[0] = W; p->Hs[0] = H;
    for (int l = 1; l < n; l++) { p->Ws[l] = (p->Ws[l - 1] + 1) / 2; p->Hs[l] = (p->Hs[l - 1] + 1) / 2; }   // impyramid: ceil(size/2)
    p->lv.assign(n, nullptr); p->dP0.assign(n, nullptr); p->dP1.assign(n, nullptr); p->dFlow.assign(n, nullptr);
    fsgm_status st = FSGM_OK;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    for (int l = 0; l < n && st == FSGM_OK && e == hipSuccess; l++) {
        // the coarsest level starts from a zero map of its own size (:34); every other level gets
        // 2*imresize(mv, 2, 'nearest') of the level above, twice that level's size (:72)
        const int mvW = l == n - 1 ? p->Ws[l] : 2 * p->Ws[l + 1], mvH = l == n - 1 ? p->Hs[l] : 2 * p->Hs[l + 1];
        st = fsgm_pyd_plan_create(&p->lv[l], p->Ws[l], p->Hs[l], mvW, mvH, prm->horSearchHalfWinSize,
                                  prm->verSearchHalfWinSize, prm->aggHalfWinSize, batch, prm->device);
        if (st != FSGM_OK) break;
        fsgm_pyd_plan* q = p->lv[l];
        (void)hipStreamDestroy(q->stream);           // all levels run on the pyramid's stream, in order
        q->stream = p->stream; q->owns_stream = false;
        st = fsgm_pyd_plan_set_params(q, prm->P1, prm->P2, prm->enableDiagonal, prm->totalPass, prm->adaptiveP2, l == 0);   // :49
        const size_t np = (size_t)p->Ws[l] * p->Hs[l];
        if (channels == 3) {
            e = hipMalloc((void**)&p->dP0[l], (size_t)batch * 3 * np);
            if (e == hipSuccess) e = hipMalloc((void**)&p->dP1[l], (size_t)batch * 3 * np);
        }
        if (e == hipSuccess) e = hipMalloc((void**)&p->dFlow[l], (size_t)batch * 2 * np * sizeof(double));
    }
    if (st == FSGM_OK && e == hipSuccess)
        e = hipMemset(p->lv[n - 1]->dMv, 0, (size_t)batch * 2 * p->lv[n - 1]->MV * sizeof(double));      // :34
    if (st != FSGM_OK || e != hipSuccess) {
        char msg[512];
        snprintf(msg, sizeof msg, "%s", st != FSGM_OK ? fsgm_last_error() : hipGetErrorString(e));
        fsgm_pyramid_plan_destroy(p);
        return fail(st != FSGM_OK ? st : (e == hipErrorOutOfMemory ? FSGM_ERR_NOMEM : FSGM_ERR_HIP), "fsgm_pyramid_plan_create: %s", msg);
    }
    *out = p;
    return FSGM_OK;
}

fsgm_status fsgm_pyramid_plan_level_size(fsgm_pyramid_plan* p, int32_t level, int32_t* w, int32_t* h) {
    FSGM_REQUIRE(p && w && h, "fsgm_pyramid_plan_level_size: null argument");
    FSGM_REQUIRE(level >= 1 && level <= p->prm.numPyd, "level %d out of range 1..%d", level, p->prm.numPyd);
    *w = p->Ws[level - 1]; *h = p->Hs[level - 1];
    return FSGM_OK;
}

fsgm_status fsgm_pyramid_plan_upload(fsgm_pyramid_plan* p, const uint8_t* I0, const uint8_t* I1) {
    return fsgm_pyramid_plan_upload_frame(p, 0, I0, I1);
}

fsgm_status fsgm_pyramid_plan_upload_frame(fsgm_pyramid_plan* p, int32_t frame, const uint8_t* I0, const uint8_t* I1) {
    FSGM_REQUIRE(p && I0 && I1, "fsgm_pyramid_plan_upload: null argument");
    FSGM_REQUIRE(frame >= 0 && frame < p->batch, "frame %d out of range (batch %d)", frame, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    const size_t n = (size_t)p->channels * p->W * p->H;
    uint8_t* d0 = (p->channels == 3 ? p->dP0[0] : p->lv[0]->dI1) + (size_t)frame * n;
    uint8_t* d1 = (p->channels == 3 ? p->dP1[0] : p->lv[0]->dI2) + (size_t)frame * n;
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    FSGM_HIP(hipMemcpyAsync(d0, I0, n, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(d1, I1, n, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

static void pyramid_enqueue_images(fsgm_pyramid_plan* p) {
    const int n = p->prm.numPyd, ch = p->channels;
    for (int l = 1; l < n; l++) {                                                // :28-31
        const uint8_t* s0 = ch == 3 ? p->dP0[l - 1] : p->lv[l - 1]->dI1;
        const uint8_t* s1 = ch == 3 ? p->dP1[l - 1] : p->lv[l - 1]->dI2;
        launch_pyr_reduce(p->stream, s0, ch == 3 ? p->dP0[l] : p->lv[l]->dI1, p->Ws[l - 1], p->Hs[l - 1], ch * p->batch);
        launch_pyr_reduce(p->stream, s1, ch == 3 ? p->dP1[l] : p->lv[l]->dI2, p->Ws[l - 1], p->Hs[l - 1], ch * p->batch);
    }
    if (ch == 3)
        for (int l = 0; l < n; l++) {                                            // :44-45
            launch_pyr_gray(p->stream, p->dP0[l], p->lv[l]->dI1, p->Ws[l], p->Hs[l], p->batch);
            launch_pyr_gray(p->stream, p->dP1[l], p->lv[l]->dI2, p->Ws[l], p->Hs[l], p->batch);
        }
}

static fsgm_status pyramid_enqueue(fsgm_pyramid_plan* p) {
    const int n = p->prm.numPyd;
    pyramid_enqueue_images(p);
    for (int l = n - 1; l >= 0; l--) {                                           // :37
        fsgm_pyd_plan* q = p->lv[l];
        fsgm_status st = pyd_enqueue(q, FSGM_STAGE_ALL, nullptr);                // :50
        if (st != FSGM_OK) return st;
        PyrFlowArgs a;
        a.bestD = q->dBestD; a.mvSub = q->dMvSub; a.mvPre = q->dMv; a.flow = p->dFlow[l];
        a.next = l > 0 ? p->lv[l - 1]->dMv : nullptr;
        a.next_frame_stride = l > 0 ? 2 * p->lv[l - 1]->MV : 0;
        a.W = q->W; a.H = q->H; a.mvW = q->mvW; a.mvH = q->mvH;
        a.Sy = q->Sy; a.hor = p->prm.horSearchHalfWinSize; a.ver = p->prm.verSearchHalfWinSize;
        launch_pyr_flow(p->stream, a, p->batch);                                 // :57-72
    }
    FSGM_HIP(hipGetLastError());
    return FSGM_OK;
}

fsgm_status fsgm_pyramid_plan_run(fsgm_pyramid_plan* p) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_HIP(hipSetDevice(p->device));
    return pyramid_enqueue(p);
}

fsgm_status fsgm_pyramid_plan_sync(fsgm_pyramid_plan* p) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_HIP(hipSetDevice(p->device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    return FSGM_OK;
}

fsgm_status fsgm_pyramid_plan_run_images(fsgm_pyramid_plan* p) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_HIP(hipSetDevice(p->device));
    pyramid_enqueue_images(p);
    FSGM_HIP(hipGetLastError());
    return FSGM_OK;
}

fsgm_status fsgm_pyramid_plan_download(fsgm_pyramid_plan* p, int32_t level, double* mv, uint32_t* minC) {
    return fsgm_pyramid_plan_download_frame(p, 0, level, mv, minC);
}

fsgm_status fsgm_pyramid_plan_download_frame(fsgm_pyramid_plan* p, int32_t frame, int32_t level, double* mv, uint32_t* minC) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(frame >= 0 && frame < p->batch, "frame %d out of range (batch %d)", frame, p->batch);
    FSGM_REQUIRE(level >= 1 && level <= p->prm.numPyd, "level %d out of range 1..%d", level, p->prm.numPyd);
    FSGM_HIP(hipSetDevice(p->device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    const int l = level - 1;
    const size_t np = (size_t)p->Ws[l] * p->Hs[l];
    if (mv) FSGM_HIP(hipMemcpy(mv, p->dFlow[l] + (size_t)frame * 2 * np, 2 * np * sizeof(double), hipMemcpyDeviceToHost));
    if (minC) FSGM_HIP(hipMemcpy(minC, p->lv[l]->dMinC + (size_t)frame * np, np * 4, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_pyramid_plan_download_gray_frame(fsgm_pyramid_plan* p, int32_t frame, int32_t level, uint8_t* g0, uint8_t* g1) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(level >= 1 && level <= p->prm.numPyd, "level %d out of range 1..%d", level, p->prm.numPyd);
    FSGM_REQUIRE(frame >= 0 && frame < p->batch, "frame %d out of range (batch %d)", frame, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    const fsgm_pyd_plan* q = p->lv[level - 1];
    const size_t o = (size_t)frame * q->NP;
    if (g0) FSGM_HIP(hipMemcpy(g0, q->dI1 + o, q->NP, hipMemcpyDeviceToHost));
    if (g1) FSGM_HIP(hipMemcpy(g1, q->dI2 + o, q->NP, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_pyramid_plan_download_gray(fsgm_pyramid_plan* p, int32_t level, uint8_t* g0, uint8_t* g1) {
    return fsgm_pyramid_plan_download_gray_frame(p, 0, level, g0, g1);
}

fsgm_status fsgm_pyramid_plan_time(fsgm_pyramid_plan* p, int32_t warmup, int32_t iters, float* ms_avg) {
    FSGM_REQUIRE(p && ms_avg && iters >= 1 && warmup >= 0, "fsgm_pyramid_plan_time: bad argument");
    FSGM_HIP(hipSetDevice(p->device));
    fsgm_status st;
    for (int i = 0; i < warmup; i++)
        if ((st = pyramid_enqueue(p)) != FSGM_OK) return st;
    FSGM_HIP(hipEventRecord(p->ev0, p->stream));
    for (int i = 0; i < iters; i++)
        if ((st = pyramid_enqueue(p)) != FSGM_OK) return st;
    FSGM_HIP(hipEventRecord(p->ev1, p->stream));
    FSGM_HIP(hipEventSynchronize(p->ev1));
    float ms = 0;
    FSGM_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *ms_avg = ms / iters;
    return FSGM_OK;
}

// ---- host-pointer entry point: one call = pyramidal_sgm(I0, I1, numPyd) ----
static PerDevice<std::vector<fsgm_pyramid_plan*>> g_pyr;        // cached plans per device, under that device's lock

void fsgm_pyramid_shutdown_internal(void) {
    for (int d = 0; d < FSGM_MAX_DEVICES; d++) {
        std::lock_guard<std::mutex> lk(g_pyr.mu[d]);
        for (fsgm_pyramid_plan* p : g_pyr.v[d]) fsgm_pyramid_plan_destroy(p);
        g_pyr.v[d].clear();
    }
}

fsgm_status fsgm_pyramidal_sgm_host(const uint8_t* I0, const uint8_t* I1, int32_t width, int32_t height, int32_t channels,
                                    const fsgm_pyramid_params* prm, double* mv, uint32_t* minC, double* const* mvPyd) {
    FSGM_REQUIRE(I0 && I1 && prm && mv, "fsgm_pyramidal_sgm: null argument");
    FSGM_DEVICE_SLOT(prm->device);
    std::lock_guard<std::mutex> lk(g_pyr.mu[prm->device]);
    std::vector<fsgm_pyramid_plan*>& g_pyr_cache = g_pyr.v[prm->device];
    fsgm_pyramid_plan* p = nullptr;
    for (fsgm_pyramid_plan* q : g_pyr_cache)
        if (q->W == width && q->H == height && q->channels == channels && q->batch == 1 && memcmp(&q->prm, prm, sizeof *prm) == 0) p = q;
    fsgm_status st;
    if (!p) {
        if ((st = fsgm_pyramid_plan_create(&p, width, height, channels, prm)) != FSGM_OK) return st;
        if (g_pyr_cache.size() >= 2) {
            fsgm_pyramid_plan_destroy(g_pyr_cache.front());
            g_pyr_cache.erase(g_pyr_cache.begin());
        }
        g_pyr_cache.push_back(p);
    }
    // One call = one stream-ordered sequence with a single host wait (like fsgm_calc_cost_sgm_batch_host): the image pair goes up
    // asynchronously, the level loop follows, every requested map comes down behind it.  (Round 3's form waited after the
    // upload, after the run and once per downloaded map, with blocking copies: 6.3 ms per call around 1.4 ms of kernels.)
    FSGM_HIP(hipSetDevice(p->device));
    StreamGuard guard(p->stream);                        // an early exit drains the stream: queued copies use the caller's memory
    const size_t nimg = (size_t)channels * width * height;
    FSGM_HIP(hipMemcpyAsync(channels == 3 ? p->dP0[0] : p->lv[0]->dI1, I0, nimg, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(channels == 3 ? p->dP1[0] : p->lv[0]->dI2, I1, nimg, hipMemcpyHostToDevice, p->stream));
    if ((st = pyramid_enqueue(p)) != FSGM_OK) return st;
    const size_t np1 = (size_t)width * height;
    FSGM_HIP(hipMemcpyAsync(mv, p->dFlow[0], 2 * np1 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    if (minC) FSGM_HIP(hipMemcpyAsync(minC, p->lv[0]->dMinC, np1 * 4, hipMemcpyDeviceToHost, p->stream));
    if (mvPyd)
        for (int l = 0; l < prm->numPyd; l++)
            if (mvPyd[l] && mvPyd[l] != mv)
                FSGM_HIP(hipMemcpyAsync(mvPyd[l], p->dFlow[l], 2 * (size_t)p->Ws[l] * p->Hs[l] * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

}  // extern "C"
