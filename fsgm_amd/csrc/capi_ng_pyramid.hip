// capi_ng_pyramid.hip -- C ABI for the pyramidal level loop around calc_pyd_cost_sgm_ng (include/fsgm.h).
// pyramidal_sgm.m:24-76 with the neighbour-guided MEX in place of calc_pyd_cost_sgm: level images, hint
// maps, candidate lists and flows stay in HBM; one host call uploads the image pair, one downloads a flow.
// The candidate volume (12 B per entry) and the sum volume are shared by the levels -- they run in order on
// one stream -- and sized for the finest.
#include "capi_common.h"
#include "epi_kernels.h"
#include "ng_kernels.h"
#include "pyramid_kernels.h"
#include <mutex>
#include <string.h>
#include <vector>

using namespace fsgm;

struct fsgm_ng_pyramid_plan {
    int W = 0, H = 0, channels = 1, device = 0, D = 0;
    int batch = 1;                                   // image pairs resident at once: every buffer below holds `batch` frames, frame-major
    fsgm_ng_pyramid_params prm{};
    std::vector<int> Ws, Hs;                         // level l (0-based) size
    std::vector<uint8_t*> dP0, dP1;                  // colour pyramids [3][h][w] (channels == 3 only)
    std::vector<uint8_t*> dG0, dG1;                  // gray pair per level
    std::vector<double*> dMv;                        // hint map per level: [2][mvH][mvW]
    std::vector<int> mvW, mvH;
    std::vector<double*> dFlow;                      // [2][h][w]
    std::vector<uint32_t*> dMinC;                    // [h][w]
    uint32_t *dCen1 = nullptr, *dCen2 = nullptr, *dS = nullptr, *dUnsafe = nullptr;
    uint16_t* dDd = nullptr;                         // repeats in the candidate lists (launch_ng_dedupe)
    uint8_t* dDk = nullptr;
    uint32_t* dBox = nullptr;                        // bounding boxes of the lists' motion vectors (grid matcher)
    uint32_t* dKstat = nullptr;                      // partial sums of the list lengths (choice of the matcher form)
    int16_t* dL4 = nullptr;                          // the compact matcher's per-path costs (ng_kernels.h)
    uint32_t* dCk = nullptr;                         // the kept entries in place order (compact aggregation kernel)
    uint16_t* dCm = nullptr;
    Cand* dC = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

extern "C" {

fsgm_ng_pyramid_params fsgm_ng_pyramid_params_default(void) {
    fsgm_ng_pyramid_params p;
    p.numPyd = 3;                      // test_psgm.m:33
    p.P1 = 6; p.P2 = 32;               // ng_sgm.m:7-8
    p.halfSearchWinSize = 1;           // ng_sgm.m:20
    p.aggSize = 2;
    p.subPixelRefine = 0;
    p.device = 0;
    return p;
}

void fsgm_ng_pyramid_plan_destroy(fsgm_ng_pyramid_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    auto drop = [](auto& v) { for (auto* b : v) if (b) (void)hipFree(b); };
    drop(p->dP0); drop(p->dP1); drop(p->dG0); drop(p->dG1); drop(p->dMv); drop(p->dFlow); drop(p->dMinC);
    void* one[] = {p->dCen1, p->dCen2, p->dS, p->dUnsafe, p->dC, p->dDd, p->dDk, p->dBox, p->dKstat, p->dCk, p->dCm, p->dL4};
    for (void* b : one) if (b) (void)hipFree(b);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

fsgm_status fsgm_ng_pyramid_plan_create(fsgm_ng_pyramid_plan** out, int32_t W, int32_t H, int32_t channels,
                                        const fsgm_ng_pyramid_params* prm) {
    return fsgm_ng_pyramid_plan_create_batch(out, W, H, channels, 1, prm);
}

fsgm_status fsgm_ng_pyramid_plan_create_batch(fsgm_ng_pyramid_plan** out, int32_t W, int32_t H, int32_t channels, int32_t batch,
                                              const fsgm_ng_pyramid_params* prm) {
    FSGM_REQUIRE(out, "fsgm_ng_pyramid_plan_create: null plan pointer");
    *out = nullptr;
    FSGM_REQUIRE(batch >= 1 && batch <= 1024, "batch must be in 1..1024 (got %d)", batch);
    FSGM_REQUIRE(prm, "fsgm_ng_pyramid_plan_create: null parameters");
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1 (got %d x %d)", W, H);
    FSGM_REQUIRE(channels == 1 || channels == 3, "channels must be 1 (gray) or 3 (RGB planes), got %d", channels);
    FSGM_REQUIRE(prm->numPyd >= 1 && prm->numPyd <= 16, "numPyd must be in 1..16 (got %d)", prm->numPyd);
    FSGM_REQUIRE(prm->halfSearchWinSize >= 0 && prm->aggSize >= 0, "halfSearchWinSize and aggSize must be >= 0");
    const long long D = 9LL * (2 * prm->halfSearchWinSize + 1) * (2 * prm->halfSearchWinSize + 1);
    if (D > FSGM_NG_MAX_D) return fail(FSGM_ERR_UNSUPPORTED, "%lld candidates per pixel exceed %d", D, FSGM_NG_MAX_D);
    if ((double)W * H * D >= 2147483648.0) return fail(FSGM_ERR_UNSUPPORTED, "candidate volume exceeds 2^31 entries");
    const size_t B = (size_t)batch;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FSGM_ERR_HIP, "no HIP device available (libfsgm_hip has no CPU fallback)");
    FSGM_REQUIRE(prm->device >= 0 && prm->device < ndev, "device %d out of range (have %d)", prm->device, ndev);
    FSGM_HIP(hipSetDevice(prm->device));
    fsgm_ng_pyramid_plan* p = new fsgm_ng_pyramid_plan;
    p->W = W; p->H = H; p->channels = channels; p->device = prm->device; p->prm = *prm; p->D = (int)D; p->batch = batch;
    const int n = prm->numPyd;
    p->Ws.resize(n); p->Hs.resize(n); p->mvW.resize(n); p->mvH.resize(n);
    p->Ws[0] = W; p->Hs[0] = H;
    for (int l = 1; l < n; l++) { p->Ws[l] = (p->Ws[l - 1] + 1) / 2; p->Hs[l] = (p->Hs[l - 1] + 1) / 2; }   // impyramid: ceil(size/2)
    p->dP0.assign(n, nullptr); p->dP1.assign(n, nullptr); p->dG0.assign(n, nullptr); p->dG1.assign(n, nullptr);
    p->dMv.assign(n, nullptr); p->dFlow.assign(n, nullptr); p->dMinC.assign(n, nullptr);
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    for (int l = 0; l < n && e == hipSuccess; l++) {
        // the coarsest level starts from a zero map of its own size (pyramidal_sgm.m:34); every other level gets
        // 2*imresize(flow, 2, 'nearest') of the level above, twice that level's size (:72)
        p->mvW[l] = l == n - 1 ? p->Ws[l] : 2 * p->Ws[l + 1];
        p->mvH[l] = l == n - 1 ? p->Hs[l] : 2 * p->Hs[l + 1];
        const size_t np = (size_t)p->Ws[l] * p->Hs[l], mv = (size_t)p->mvW[l] * p->mvH[l];
        if (channels == 3) {
            e = hipMalloc((void**)&p->dP0[l], B * 3 * np);
            if (e == hipSuccess) e = hipMalloc((void**)&p->dP1[l], B * 3 * np);
        }
        if (e == hipSuccess) e = hipMalloc((void**)&p->dG0[l], B * np);
        if (e == hipSuccess) e = hipMalloc((void**)&p->dG1[l], B * np);
        if (e == hipSuccess) e = hipMalloc((void**)&p->dMv[l], B * 2 * mv * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&p->dFlow[l], B * 2 * np * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&p->dMinC[l], B * np * 4);
    }
    const size_t NP = (size_t)W * H, N = NP * (size_t)D;
    if (e == hipSuccess) e = hipMalloc((void**)&p->dCen1, B * NP * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->dCen2, B * NP * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->dC, B * N * sizeof(Cand));
    if (e == hipSuccess) e = hipMalloc((void**)&p->dS, B * N * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->dUnsafe, 8);
    if (e == hipSuccess) e = hipMalloc((void**)&p->dDd, B * N * 2);
    if (e == hipSuccess) e = hipMalloc((void**)&p->dDk, B * NP);
    if (e == hipSuccess) e = hipMalloc((void**)&p->dBox, B * NP * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->dKstat, NG_KSTAT_WORDS * 4);
    if (e == hipSuccess && p->D <= 128) e = hipMalloc((void**)&p->dL4, (size_t)B * p->Ws[0] * p->Hs[0] * NG_L4_PER_PIXEL * sizeof(int16_t));
    if (e == hipSuccess) e = hipMalloc((void**)&p->dCk, B * N * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->dCm, B * N * 2);
    if (e == hipSuccess) e = hipMemset(p->dMv[n - 1], 0, B * 2 * (size_t)p->mvW[n - 1] * p->mvH[n - 1] * sizeof(double));   // :34
    if (e != hipSuccess) {
        fsgm_ng_pyramid_plan_destroy(p);
        return fail(e == hipErrorOutOfMemory ? FSGM_ERR_NOMEM : FSGM_ERR_HIP, "fsgm_ng_pyramid_plan_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return FSGM_OK;
}

fsgm_status fsgm_ng_pyramid_plan_level_size(fsgm_ng_pyramid_plan* p, int32_t level, int32_t* w, int32_t* h) {
    FSGM_REQUIRE(p && w && h, "fsgm_ng_pyramid_plan_level_size: null argument");
    FSGM_REQUIRE(level >= 1 && level <= p->prm.numPyd, "level %d out of range 1..%d", level, p->prm.numPyd);
    *w = p->Ws[level - 1]; *h = p->Hs[level - 1];
    return FSGM_OK;
}

fsgm_status fsgm_ng_pyramid_plan_upload_frame(fsgm_ng_pyramid_plan* p, int32_t frame, const uint8_t* I0, const uint8_t* I1) {
    FSGM_REQUIRE(p && I0 && I1, "fsgm_ng_pyramid_plan_upload: null argument");
    FSGM_REQUIRE(frame >= 0 && frame < p->batch, "frame %d out of range (batch %d)", frame, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    const size_t n = (size_t)p->channels * p->W * p->H;
    StreamGuard guard(p->stream);
    FSGM_HIP(hipMemcpyAsync((p->channels == 3 ? p->dP0[0] : p->dG0[0]) + frame * n, I0, n, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync((p->channels == 3 ? p->dP1[0] : p->dG1[0]) + frame * n, I1, n, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_ng_pyramid_plan_upload(fsgm_ng_pyramid_plan* p, const uint8_t* I0, const uint8_t* I1) {
    return fsgm_ng_pyramid_plan_upload_frame(p, 0, I0, I1);
}

static fsgm_status ng_pyramid_enqueue(fsgm_ng_pyramid_plan* p) {
    const int n = p->prm.numPyd, ch = p->channels, D = p->D, B = p->batch;
    hipStream_t s = p->stream;
    for (int l = 1; l < n; l++) {                                                // pyramidal_sgm.m:28-31 (all frames' planes in one launch)
        launch_pyr_reduce(s, ch == 3 ? p->dP0[l - 1] : p->dG0[l - 1], ch == 3 ? p->dP0[l] : p->dG0[l], p->Ws[l - 1], p->Hs[l - 1], ch * B);
        launch_pyr_reduce(s, ch == 3 ? p->dP1[l - 1] : p->dG1[l - 1], ch == 3 ? p->dP1[l] : p->dG1[l], p->Ws[l - 1], p->Hs[l - 1], ch * B);
    }
    if (ch == 3)
        for (int l = 0; l < n; l++) {                                            // :44-45
            launch_pyr_gray(s, p->dP0[l], p->dG0[l], p->Ws[l], p->Hs[l], B);          // all frames in one launch
            launch_pyr_gray(s, p->dP1[l], p->dG1[l], p->Ws[l], p->Hs[l], B);
        }
    for (int l = n - 1; l >= 0; l--) {                                           // :37
        const int w = p->Ws[l], h = p->Hs[l];
        const size_t N = (size_t)w * h * D;
        // 4-byte candidate entries (3x3 hint kernel, D = 81): the keys live in S's memory until the matchers need S (ng_kernels.h)
        static const bool k4_env = [] { const char* e = getenv("FSGM_NG_K4"); return !(e && e[0] == '0'); }();
        static const bool hint_env = [] { const char* e = getenv("FSGM_NG_COST_HINT"); return !(e && e[0] == '0'); }();
        const bool dedupe_on = [] { const char* e = getenv("FSGM_NG_DEDUPE"); return !(e && atoi(e) == 0); }();
        const bool k4 = k4_env && hint_env && dedupe_on && p->prm.halfSearchWinSize == 1 && p->prm.aggSize / 2 == 1 && D <= 128;
        if (!k4) FSGM_HIP(hipMemsetAsync(p->dS, 0, (size_t)B * N * 4, s));       // calc_pyd_cost_sgm_ng.cpp:111
        FSGM_HIP(hipMemsetAsync(p->dUnsafe, 0, 8, s));
        launch_census(s, p->dG0[l], p->dCen1, w, h, B);                          // :485-486
        launch_census(s, p->dG1[l], p->dCen2, w, h, B);
        NgCostArgs ca;
        ca.K4 = k4 ? p->dS : nullptr; ca.flags = k4 ? p->dUnsafe : nullptr;
        ca.cen1 = p->dCen1; ca.cen2 = p->dCen2; ca.mv = p->dMv[l]; ca.C = p->dC; ca.unsafe = p->dUnsafe; ca.W = w; ca.H = h;
        ca.mvW = p->mvW[l]; ca.mvH = p->mvH[l]; ca.rAgg = p->prm.aggSize / 2; ca.rX = p->prm.halfSearchWinSize; ca.rY = p->prm.halfSearchWinSize;
        launch_ng_cost(s, ca, B);
        NgAggArgs ga;
        ga.C = p->dC; ga.S = p->dS; ga.unsafe = p->dUnsafe; ga.W = w; ga.H = h; ga.D = D; ga.P1 = p->prm.P1; ga.P2 = p->prm.P2;
        ga.dd = nullptr; ga.dk = nullptr; ga.dbox = nullptr; ga.kstat = nullptr; ga.ck = nullptr; ga.cm = nullptr; ga.L4 = nullptr;
        if (D <= 128) {
            launch_ng_dedupe(s, p->dC, p->dDd, p->dDk, p->dBox, p->dKstat, p->dCk, p->dCm, w, h, D, B, ca.K4, ca.flags);
            ga.dd = p->dDd; ga.dk = p->dDk; ga.dbox = p->dBox; ga.kstat = p->dKstat; ga.ck = p->dCk; ga.cm = p->dCm; ga.L4 = p->dL4;
            if (k4) launch_ng_prepare_matchers(s, ga, p->dS, p->dC, p->dUnsafe, B);
        }
        launch_ng_aggregate(s, ga, B);
        NgWtaArgs wa;
        wa.C = p->dC; wa.S = p->dS; wa.minC = p->dMinC[l]; wa.flow = p->dFlow[l]; wa.W = w; wa.H = h; wa.D = D;
        wa.cm = ga.dd ? p->dCm : nullptr; wa.dk = ga.dd ? p->dDk : nullptr; wa.L4 = ga.L4; wa.kstat = ga.kstat; wa.K4 = ca.K4; wa.flags = ca.flags;
        launch_ng_wta(s, wa, B);
        if (p->prm.subPixelRefine) {                                             // :516-517
            NgSubpixArgs sa;
            sa.cen1 = p->dCen1; sa.cen2 = p->dCen2; sa.flow = p->dFlow[l]; sa.W = w; sa.H = h;
            launch_ng_subpixel(s, sa, B);
        }
        if (l > 0)                                                               // pyramidal_sgm.m:72
            launch_pyr_upsample2(s, p->dFlow[l], p->dMv[l - 1], w, h, B, (size_t)2 * p->mvW[l - 1] * p->mvH[l - 1]);
    }
    FSGM_HIP(hipGetLastError());
    return FSGM_OK;
}

fsgm_status fsgm_ng_pyramid_plan_run(fsgm_ng_pyramid_plan* p) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_HIP(hipSetDevice(p->device));
    return ng_pyramid_enqueue(p);
}

fsgm_status fsgm_ng_pyramid_plan_download_frame(fsgm_ng_pyramid_plan* p, int32_t frame, int32_t level, double* flow, uint32_t* minC) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(level >= 1 && level <= p->prm.numPyd, "level %d out of range 1..%d", level, p->prm.numPyd);
    FSGM_REQUIRE(frame >= 0 && frame < p->batch, "frame %d out of range (batch %d)", frame, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    const int l = level - 1;
    const size_t np = (size_t)p->Ws[l] * p->Hs[l];
    if (flow) FSGM_HIP(hipMemcpy(flow, p->dFlow[l] + (size_t)frame * 2 * np, 2 * np * sizeof(double), hipMemcpyDeviceToHost));
    if (minC) FSGM_HIP(hipMemcpy(minC, p->dMinC[l] + (size_t)frame * np, np * 4, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_ng_pyramid_plan_download(fsgm_ng_pyramid_plan* p, int32_t level, double* flow, uint32_t* minC) {
    return fsgm_ng_pyramid_plan_download_frame(p, 0, level, flow, minC);
}

fsgm_status fsgm_ng_pyramid_plan_time(fsgm_ng_pyramid_plan* p, int32_t warmup, int32_t iters, float* ms_avg) {
    FSGM_REQUIRE(p && ms_avg && iters >= 1 && warmup >= 0, "fsgm_ng_pyramid_plan_time: bad argument");
    FSGM_HIP(hipSetDevice(p->device));
    fsgm_status st;
    for (int i = 0; i < warmup; i++)
        if ((st = ng_pyramid_enqueue(p)) != FSGM_OK) return st;
    FSGM_HIP(hipEventRecord(p->ev0, p->stream));
    for (int i = 0; i < iters; i++)
        if ((st = ng_pyramid_enqueue(p)) != FSGM_OK) return st;
    FSGM_HIP(hipEventRecord(p->ev1, p->stream));
    FSGM_HIP(hipEventSynchronize(p->ev1));
    float ms = 0;
    FSGM_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *ms_avg = ms / iters;
    return FSGM_OK;
}

// ---- host-pointer entry point: one call = the whole loop; plans are cached per shape like fsgm_pyramidal_sgm_host's ----
static PerDevice<std::vector<fsgm_ng_pyramid_plan*>> g_ngpyr;   // cached plans per device, under that device's lock

void fsgm_ng_pyramid_shutdown_internal(void) {
    for (int d = 0; d < FSGM_MAX_DEVICES; d++) {
        std::lock_guard<std::mutex> lk(g_ngpyr.mu[d]);
        for (fsgm_ng_pyramid_plan* p : g_ngpyr.v[d]) fsgm_ng_pyramid_plan_destroy(p);
        g_ngpyr.v[d].clear();
    }
}

fsgm_status fsgm_pyramidal_sgm_ng_host(const uint8_t* I0, const uint8_t* I1, int32_t width, int32_t height, int32_t channels,
                                       const fsgm_ng_pyramid_params* prm, double* flow, uint32_t* minC, double* const* flowPyd) {
    FSGM_REQUIRE(I0 && I1 && prm && flow, "fsgm_pyramidal_sgm_ng: null argument");
    FSGM_DEVICE_SLOT(prm->device);
    std::lock_guard<std::mutex> lk(g_ngpyr.mu[prm->device]);
    std::vector<fsgm_ng_pyramid_plan*>& g_ngpyr_cache = g_ngpyr.v[prm->device];
    fsgm_ng_pyramid_plan* p = nullptr;
    for (fsgm_ng_pyramid_plan* q : g_ngpyr_cache)
        if (q->W == width && q->H == height && q->channels == channels && q->batch == 1 && memcmp(&q->prm, prm, sizeof *prm) == 0) p = q;
    fsgm_status st;
    if (!p) {
        if ((st = fsgm_ng_pyramid_plan_create(&p, width, height, channels, prm)) != FSGM_OK) return st;
        if (g_ngpyr_cache.size() >= 2) {
            fsgm_ng_pyramid_plan_destroy(g_ngpyr_cache.front());
            g_ngpyr_cache.erase(g_ngpyr_cache.begin());
        }
        g_ngpyr_cache.push_back(p);
    }
    // one stream-ordered sequence, a single host wait (see fsgm_pyramidal_sgm_host)
    FSGM_HIP(hipSetDevice(p->device));
    StreamGuard guard(p->stream);
    const size_t nimg = (size_t)channels * width * height;
    FSGM_HIP(hipMemcpyAsync(channels == 3 ? p->dP0[0] : p->dG0[0], I0, nimg, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(channels == 3 ? p->dP1[0] : p->dG1[0], I1, nimg, hipMemcpyHostToDevice, p->stream));
    if ((st = ng_pyramid_enqueue(p)) != FSGM_OK) return st;
    const size_t np1 = (size_t)width * height;
    FSGM_HIP(hipMemcpyAsync(flow, p->dFlow[0], 2 * np1 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    if (minC) FSGM_HIP(hipMemcpyAsync(minC, p->dMinC[0], np1 * 4, hipMemcpyDeviceToHost, p->stream));
    if (flowPyd)
        for (int l = 0; l < prm->numPyd; l++)
            if (flowPyd[l] && flowPyd[l] != flow)
                FSGM_HIP(hipMemcpyAsync(flowPyd[l], p->dFlow[l], 2 * (size_t)p->Ws[l] * p->Hs[l] * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

}  // extern "C"
