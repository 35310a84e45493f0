// capi_pyd.hip -- C ABI for the calc_pyd_cost_sgm path (include/fsgm.h).
#include "capi_common.h"
#include "epi_kernels.h"
#include "pyd_kernels.h"
#include "pyd_plan.h"
#include <mutex>
#include <string.h>
#include <stdlib.h>
#include <vector>

using namespace fsgm;


extern "C" {

void fsgm_pyd_plan_destroy(fsgm_pyd_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    void* bufs[] = {p->dI1, p->dI2, p->dC, p->dL, p->dCen1, p->dCen2, p->dBestD, p->dMinC, p->dS, p->dDesc, p->dMv, p->dMvSub};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream && p->owns_stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

fsgm_status fsgm_pyd_plan_create(fsgm_pyd_plan** out, int32_t W, int32_t H, int32_t mvW, int32_t mvH,
                                 int32_t rX, int32_t rY, int32_t rAgg, int32_t batch, int32_t device) {
    FSGM_REQUIRE(out, "fsgm_pyd_plan_create: null plan pointer");
    *out = nullptr;
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1 (got %d x %d)", W, H);
    FSGM_REQUIRE(mvW >= W && mvH >= H, "preMv (%d x %d) must be at least as large as the image (%d x %d): the reference "
                 "indexes it with image coordinates (calc_pyd_cost_sgm.cpp:388-389)", mvW, mvH, W, H);
    FSGM_REQUIRE(rX >= 0 && rY >= 0 && rAgg >= 0, "window half sizes must be >= 0");
    if (2 * rX + 1 > FSGM_PYD_MAX_SIDE || 2 * rY + 1 > FSGM_PYD_MAX_SIDE)
        return fail(FSGM_ERR_UNSUPPORTED, "search window side exceeds %d", FSGM_PYD_MAX_SIDE);
    FSGM_REQUIRE(batch >= 1, "batch must be >= 1");
    const long long D = (long long)(2 * rX + 1) * (2 * rY + 1);
    if (D > FSGM_PYD_MAX_D) return fail(FSGM_ERR_UNSUPPORTED, "search window %lld candidates exceeds %d", D, FSGM_PYD_MAX_D);
    if ((double)W * H * D >= 2147483648.0) return fail(FSGM_ERR_UNSUPPORTED, "cost volume exceeds 2^31 voxels per frame");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FSGM_ERR_HIP, "no HIP device available (libfsgm_hip has no CPU fallback)");
    FSGM_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
    FSGM_HIP(hipSetDevice(device));
    fsgm_pyd_plan* p = new fsgm_pyd_plan;
    p->W = W; p->H = H; p->mvW = mvW; p->mvH = mvH; p->rX = rX; p->rY = rY; p->rAgg = rAgg;
    p->batch = batch; p->device = device;
    p->Sx = 2 * rX + 1; p->Sy = 2 * rY + 1; p->D = (int)D;
    p->RS = pyd_row_stride(p->Sx, p->Sy); p->PS = p->Sx * p->RS;
    p->NP = (size_t)W * H; p->N = p->NP * p->PS; p->MV = (size_t)mvW * mvH;
    const size_t B = batch;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** ptr, size_t bytes) { if (e == hipSuccess) e = hipMalloc(ptr, bytes); };
    alloc((void**)&p->dI1, B * p->NP);
    alloc((void**)&p->dI2, B * p->NP);
    alloc((void**)&p->dCen1, B * p->NP * 4);
    alloc((void**)&p->dCen2, B * p->NP * 4);
    alloc((void**)&p->dMv, B * p->MV * 16);
    alloc((void**)&p->dC, B * p->N);
    alloc((void**)&p->dL, B * p->N * 8);
    // (the row-packed aggregation addresses a frame's volume with 32-bit byte offsets: larger frames take the generic kernels)
    if (pyd_rows_layout(p->Sx, p->Sy) && p->N < (1ull << 32)) alloc((void**)&p->dDesc, (B * p->NP * 8 + 4) * 4);   // + the dump slot
    alloc((void**)&p->dBestD, B * p->NP * 4);
    alloc((void**)&p->dMinC, B * p->NP * 4);
    alloc((void**)&p->dMvSub, B * p->NP * 16);
    if (e == hipSuccess) e = hipMemset(p->dC, 0, B * p->N);         // padding bytes of the rows layout stay defined
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    if (e != hipSuccess) {
        fsgm_pyd_plan_destroy(p);
        return fail(e == hipErrorOutOfMemory ? FSGM_ERR_NOMEM : FSGM_ERR_HIP, "fsgm_pyd_plan_create: %s", hipGetErrorString(e));
    }
    // once per device: the row-packed aggregation's 3-input minima (v_pk_minimum3_f16 on denormal patterns, epi_sweep.hip's self-test)
    // must be exact u16 operations; where they are not, the plan has no descriptors and the generic kernels run
    if (p->dDesc) {
        static std::mutex mu;
        static int state[64] = {0};                          // 0 unknown, 1 good, 2 bad
        std::lock_guard<std::mutex> lk(mu);
        int& st = state[device & 63];
        if (st == 0) st = fused_step_selftest(p->stream) == 0 ? 1 : 2;
        if (st == 2) { (void)hipFree(p->dDesc); p->dDesc = nullptr; }
    }
    *out = p;
    return FSGM_OK;
}

fsgm_status fsgm_pyd_plan_set_params(fsgm_pyd_plan* p, int32_t P1, int32_t P2, int32_t diag, int32_t totalPass,
                                     int32_t adaptive, int32_t subpixel) {
    FSGM_REQUIRE(p, "null plan");
    p->P1 = P1; p->P2 = P2; p->diagonal = diag != 0; p->totalPass = totalPass;
    p->adaptive = adaptive != 0; p->subpixel = subpixel;
    return FSGM_OK;
}

fsgm_status fsgm_pyd_plan_upload(fsgm_pyd_plan* p, int32_t f, const uint8_t* I1, const uint8_t* I2, const double* mv) {
    FSGM_REQUIRE(p && I1 && I2 && mv, "fsgm_pyd_plan_upload: null argument");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    FSGM_HIP(hipMemcpyAsync(p->dI1 + f * p->NP, I1, p->NP, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(p->dI2 + f * p->NP, I2, p->NP, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(p->dMv + f * 2 * p->MV, mv, p->MV * 16, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_pyd_plan_upload_cost(fsgm_pyd_plan* p, int32_t f, const uint8_t* C) {
    FSGM_REQUIRE(p && C, "fsgm_pyd_plan_upload_cost: null argument");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    int cm = p->cmax;
    const size_t ND = p->NP * p->D;
    for (size_t i = 0; i < ND; i++) cm = C[i] > cm ? C[i] : cm;
    p->cmax = cm;
    const uint8_t* src = C;
    if (p->PS != p->D) {                 // reference order [pixel][sx*Sy+sy] -> padded rows
        p->stage.assign(p->N, 0);
        for (size_t px = 0; px < p->NP; px++)
            for (int sx = 0; sx < p->Sx; sx++)
                memcpy(&p->stage[px * p->PS + (size_t)sx * p->RS], C + px * p->D + (size_t)sx * p->Sy, p->Sy);
        src = p->stage.data();
    }
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    FSGM_HIP(hipMemcpyAsync(p->dC + f * p->N, src, p->N, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

}  // extern "C"

// FSGM_PYD_WIDE: 0 = never use the wide mapping, 1 = automatic (default), 2 = always (tests)
static int fsgm_env_wide() {
    const char* e = getenv("FSGM_PYD_WIDE");
    return (e && *e) ? atoi(e) : 1;
}

fsgm_status fsgm::pyd_enqueue(fsgm_pyd_plan* p, int stages, uint32_t* dS) {
    PydAggArgs g;
    PydWtaArgs w;
    g.I1 = p->dI1; g.C = p->dC; g.mv = p->dMv; g.L = p->dL; g.desc = p->dDesc; g.dump = p->dDesc ? p->dDesc + (size_t)p->batch * p->NP * 8 : nullptr;
    g.W = p->W; g.H = p->H; g.mvW = p->mvW; g.mvH = p->mvH; g.Sx = p->Sx; g.Sy = p->Sy;
    g.RS = p->RS; g.PS = p->PS;
    g.P1 = p->P1; g.P2 = p->P2; g.adaptive = p->adaptive;
    // cmax: 24 for volumes built here (census 5x5 Hamming mean; out-of-image taps add 5), else as uploaded
    const int cm = (stages & FSGM_STAGE_COST) ? 24 : p->cmax;
    const bool nowrap = p->P1 >= 0 && p->P2 >= 0 && cm + p->P2 + (p->P1 > p->P2 ? p->P1 : p->P2) <= 255;
    const bool rows = nowrap && p->dDesc != nullptr;        // row-packed aggregation (pyd_rows.hip)
    // A single landscape frame is bounded by its longest serial chains, the horizontal lines: those take
    // the kernel's one-line-per-wave mapping (fewer instructions per step); batches keep the packed one.
    const bool wide_rows = rows && p->batch <= 2 && p->W > p->H && fsgm_env_wide() != 0;
    plan_pyd_dirs(g, p->diagonal, p->totalPass, w.weight, rows ? 16 : 4, wide_rows || (rows && fsgm_env_wide() == 2));
    if (stages & FSGM_STAGE_COST) {
        launch_census(p->stream, p->dI1, p->dCen1, p->W, p->H, p->batch);       // :485-486
        launch_census(p->stream, p->dI2, p->dCen2, p->W, p->H, p->batch);
        PydCostArgs a;
        a.cen1 = p->dCen1; a.cen2 = p->dCen2; a.mv = p->dMv; a.C = p->dC;
        a.W = p->W; a.H = p->H; a.mvW = p->mvW; a.mvH = p->mvH; a.rAgg = p->rAgg; a.rX = p->rX; a.rY = p->rY;
        a.RS = p->RS; a.PS = p->PS;
        launch_pyd_cost(p->stream, a, p->batch);
        p->cmax = 24;
    }
    if (stages & FSGM_STAGE_AGGREGATE) {
        if (rows) {
            // (making the descriptors on a side stream while the cost volume is built was tried: the event
            // fork/join costs more than the 0.06 ms it hides -- 1.71 vs 1.68 ms per 3-level pyramid)
            launch_pyd_rows_desc(p->stream, g, p->batch);
            launch_pyd_rows_aggregate(p->stream, g, p->batch);
        } else {
            launch_pyd_aggregate(p->stream, g, p->batch, !nowrap);
        }
    }
    if (stages & FSGM_STAGE_WTA) {
        w.L = p->dL; w.bestD = p->dBestD; w.minC = p->dMinC; w.mvSub = p->dMvSub; w.S = dS;
        w.W = p->W; w.H = p->H; w.Sx = p->Sx; w.Sy = p->Sy; w.RS = p->RS; w.PS = p->PS;
        w.ndirs = g.ndirs; w.subpixel = p->subpixel;
        launch_pyd_wta(p->stream, w, p->batch);
    }
    FSGM_HIP(hipGetLastError());
    return FSGM_OK;
}

extern "C" {

fsgm_status fsgm_pyd_plan_run(fsgm_pyd_plan* p, int32_t stages) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE((stages & ~FSGM_STAGE_ALL) == 0 && stages != 0, "bad stage mask %d", stages);
    FSGM_HIP(hipSetDevice(p->device));
    return pyd_enqueue(p, stages, nullptr);
}

fsgm_status fsgm_pyd_plan_download(fsgm_pyd_plan* p, int32_t f, uint32_t* bestD, uint32_t* minC, double* mvSub) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    if (bestD) FSGM_HIP(hipMemcpy(bestD, p->dBestD + f * p->NP, p->NP * 4, hipMemcpyDeviceToHost));
    if (minC) FSGM_HIP(hipMemcpy(minC, p->dMinC + f * p->NP, p->NP * 4, hipMemcpyDeviceToHost));
    if (mvSub) FSGM_HIP(hipMemcpy(mvSub, p->dMvSub + f * 2 * p->NP, p->NP * 16, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_pyd_plan_download_cost(fsgm_pyd_plan* p, int32_t f, uint8_t* C) {
    FSGM_REQUIRE(p && C, "fsgm_pyd_plan_download_cost: null argument");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    if (p->PS == p->D) {
        FSGM_HIP(hipMemcpy(C, p->dC + f * p->N, p->N, hipMemcpyDeviceToHost));
        return FSGM_OK;
    }
    p->stage.resize(p->N);               // padded rows -> reference order
    FSGM_HIP(hipMemcpy(p->stage.data(), p->dC + f * p->N, p->N, hipMemcpyDeviceToHost));
    for (size_t px = 0; px < p->NP; px++)
        for (int sx = 0; sx < p->Sx; sx++)
            memcpy(C + px * p->D + (size_t)sx * p->Sy, &p->stage[px * p->PS + (size_t)sx * p->RS], p->Sy);
    return FSGM_OK;
}

fsgm_status fsgm_pyd_plan_download_sum(fsgm_pyd_plan* p, int32_t f, uint32_t* S) {
    FSGM_REQUIRE(p && S, "fsgm_pyd_plan_download_sum: null argument");
    FSGM_REQUIRE(f >= 0 && f < p->batch, "frame %d out of range (batch %d)", f, p->batch);
    FSGM_HIP(hipSetDevice(p->device));
    const size_t ND = p->NP * p->D;
    if (!p->dS) FSGM_HIP(hipMalloc((void**)&p->dS, (size_t)p->batch * ND * 4));
    fsgm_status st = pyd_enqueue(p, FSGM_STAGE_WTA, p->dS);      // the WTA kernel taps S on its way
    if (st != FSGM_OK) return st;
    FSGM_HIP(hipStreamSynchronize(p->stream));
    FSGM_HIP(hipMemcpy(S, p->dS + f * ND, ND * 4, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_pyd_plan_time(fsgm_pyd_plan* p, int32_t stages, int32_t warmup, int32_t iters, float* ms_avg) {
    FSGM_REQUIRE(p && ms_avg && iters >= 1 && warmup >= 0, "fsgm_pyd_plan_time: bad argument");
    FSGM_REQUIRE((stages & ~FSGM_STAGE_ALL) == 0 && stages != 0, "bad stage mask %d", stages);
    FSGM_HIP(hipSetDevice(p->device));
    fsgm_status st;
    for (int i = 0; i < warmup; i++)
        if ((st = pyd_enqueue(p, stages, nullptr)) != FSGM_OK) return st;
    FSGM_HIP(hipEventRecord(p->ev0, p->stream));
    for (int i = 0; i < iters; i++)
        if ((st = pyd_enqueue(p, stages, nullptr)) != FSGM_OK) return st;
    FSGM_HIP(hipEventRecord(p->ev1, p->stream));
    FSGM_HIP(hipEventSynchronize(p->ev1));
    float ms = 0;
    FSGM_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *ms_avg = ms / iters;
    return FSGM_OK;
}

// ---- host-pointer entry points (the calc_pyd_cost_sgm gateway) ----
static PerDevice<std::vector<fsgm_pyd_plan*>> g_pyd;            // cached plans per device, under that device's lock

void fsgm_pyd_shutdown_internal(void) {
    for (int d = 0; d < FSGM_MAX_DEVICES; d++) {
        std::lock_guard<std::mutex> lk(g_pyd.mu[d]);
        for (fsgm_pyd_plan* p : g_pyd.v[d]) fsgm_pyd_plan_destroy(p);
        g_pyd.v[d].clear();
    }
}

fsgm_status fsgm_calc_pyd_cost_sgm_batch_host(int32_t n, const fsgm_pyd_in* in, const fsgm_pyd_out* out, int32_t device) {
    FSGM_REQUIRE(n >= 1 && in && out, "fsgm_calc_pyd_cost_sgm: null argument");
    const fsgm_pyd_in& a = in[0];
    for (int i = 0; i < n; i++) {
        FSGM_REQUIRE(in[i].I1 && in[i].I2 && in[i].preMv, "fsgm_calc_pyd_cost_sgm: frame %d has a null input", i);
        FSGM_REQUIRE(out[i].bestD && out[i].minC && out[i].mvSub, "fsgm_calc_pyd_cost_sgm: frame %d has a null output", i);
        const fsgm_pyd_in& b = in[i];
        FSGM_REQUIRE(b.width == a.width && b.height == a.height && b.mvWidth == a.mvWidth && b.mvHeight == a.mvHeight &&
                     b.halfSearchWinSizeX == a.halfSearchWinSizeX && b.halfSearchWinSizeY == a.halfSearchWinSizeY &&
                     b.aggHalfWinSize == a.aggHalfWinSize && b.subPixelRefine == a.subPixelRefine && b.P1 == a.P1 &&
                     b.P2 == a.P2 && b.enableDiagnalPath == a.enableDiagnalPath && b.totalPass == a.totalPass &&
                     b.adpativeP2 == a.adpativeP2,
                     "frames of one batch must share shape and parameters (frame %d differs)", i);
    }
    FSGM_DEVICE_SLOT(device);
    std::lock_guard<std::mutex> lk(g_pyd.mu[device]);
    std::vector<fsgm_pyd_plan*>& g_pyd_cache = g_pyd.v[device];
    fsgm_pyd_plan* p = nullptr;
    for (fsgm_pyd_plan* q : g_pyd_cache)
        if (q->W == a.width && q->H == a.height && q->mvW == a.mvWidth && q->mvH == a.mvHeight &&
            q->rX == a.halfSearchWinSizeX && q->rY == a.halfSearchWinSizeY && q->rAgg == a.aggHalfWinSize &&
            q->batch == n && q->device == device) p = q;
    fsgm_status st;
    if (!p) {
        st = fsgm_pyd_plan_create(&p, a.width, a.height, a.mvWidth, a.mvHeight, a.halfSearchWinSizeX,
                                  a.halfSearchWinSizeY, a.aggHalfWinSize, n, device);
        if (st != FSGM_OK) return st;
        if (g_pyd_cache.size() >= 6) {       // a pyramid visits ~5 shapes per frame pair
            fsgm_pyd_plan_destroy(g_pyd_cache.front());
            g_pyd_cache.erase(g_pyd_cache.begin());
        }
        g_pyd_cache.push_back(p);
    }
    if ((st = fsgm_pyd_plan_set_params(p, a.P1, a.P2, a.enableDiagnalPath, a.totalPass, a.adpativeP2, a.subPixelRefine)) != FSGM_OK) return st;
    for (int i = 0; i < n; i++)
        if ((st = fsgm_pyd_plan_upload(p, i, in[i].I1, in[i].I2, in[i].preMv)) != FSGM_OK) return st;
    if ((st = fsgm_pyd_plan_run(p, FSGM_STAGE_ALL)) != FSGM_OK) return st;
    for (int i = 0; i < n; i++) {
        if ((st = fsgm_pyd_plan_download(p, i, out[i].bestD, out[i].minC, out[i].mvSub)) != FSGM_OK) return st;
        if (out[i].C && (st = fsgm_pyd_plan_download_cost(p, i, out[i].C)) != FSGM_OK) return st;
        if (out[i].S && (st = fsgm_pyd_plan_download_sum(p, i, out[i].S)) != FSGM_OK) return st;
    }
    return FSGM_OK;
}

fsgm_status fsgm_calc_pyd_cost_sgm_host(const fsgm_pyd_in* in, const fsgm_pyd_out* out, int32_t device) {
    return fsgm_calc_pyd_cost_sgm_batch_host(1, in, out, device);
}

}  // extern "C"
