// capi_post.hip -- C ABI for the post-processing functions the reference chains after SGM
// (test.m:45-50): speckle_filter.m, calc_disp_from_first.m, forward_backward_check.m,
// scanline_in_fill.m, vzInd2Disp.m.  One entry point per MATLAB function (host pointers in / out) and a
// device-resident plan that runs the whole chain without leaving HBM.
#include "capi_common.h"
#include "post_kernels.h"
#include <math.h>
#include <mutex>
#include <vector>

using namespace fsgm;

struct fsgm_post_plan {
    int W = 0, H = 0, device = 0;
    size_t NP = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *dIn = nullptr, *dA = nullptr, *dB = nullptr, *dD2 = nullptr, *dOut = nullptr, *dDisp = nullptr;
    double *dPd0 = nullptr, *dNd = nullptr, *dO = nullptr;
    int32_t *dParent = nullptr, *dSize = nullptr, *dScan = nullptr, *dLabels = nullptr, *dLeft = nullptr;
};

extern "C" {

void fsgm_post_plan_destroy(fsgm_post_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    void* bufs[] = {p->dIn, p->dA, p->dB, p->dD2, p->dOut, p->dDisp, p->dPd0, p->dNd, p->dO,
                    p->dParent, p->dSize, p->dScan, p->dLabels, p->dLeft};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

fsgm_status fsgm_post_plan_create(fsgm_post_plan** out, int32_t W, int32_t H, int32_t device) {
    FSGM_REQUIRE(out, "fsgm_post_plan_create: null plan pointer");
    *out = nullptr;
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1 (got %d x %d)", W, H);
    if ((double)W * H >= 2147483648.0) return fail(FSGM_ERR_UNSUPPORTED, "map exceeds 2^31 pixels");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FSGM_ERR_HIP, "no HIP device available (libfsgm_hip has no CPU fallback)");
    FSGM_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
    FSGM_HIP(hipSetDevice(device));
    fsgm_post_plan* p = new fsgm_post_plan;
    p->W = W; p->H = H; p->device = device; p->NP = (size_t)W * H;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** ptr, size_t bytes) { if (e == hipSuccess) e = hipMalloc(ptr, bytes); };
    for (double** b : {&p->dIn, &p->dA, &p->dB, &p->dD2, &p->dOut, &p->dDisp, &p->dO}) alloc((void**)b, p->NP * 8);
    alloc((void**)&p->dPd0, p->NP * 16);
    alloc((void**)&p->dNd, p->NP * 16);
    for (int32_t** b : {&p->dParent, &p->dSize, &p->dLabels, &p->dLeft}) alloc((void**)b, p->NP * 4);
    alloc((void**)&p->dScan, (p->NP / 1024 + 2) * 4);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    if (e != hipSuccess) {
        fsgm_post_plan_destroy(p);
        return fail(e == hipErrorOutOfMemory ? FSGM_ERR_NOMEM : FSGM_ERR_HIP, "fsgm_post_plan_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return FSGM_OK;
}

fsgm_status fsgm_post_plan_upload(fsgm_post_plan* p, const double* D1, const double* Pd0, const double* normDirect, const double* O) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_HIP(hipSetDevice(p->device));
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    if (D1) FSGM_HIP(hipMemcpyAsync(p->dIn, D1, p->NP * 8, hipMemcpyHostToDevice, p->stream));
    if (Pd0) FSGM_HIP(hipMemcpyAsync(p->dPd0, Pd0, p->NP * 16, hipMemcpyHostToDevice, p->stream));
    if (normDirect) FSGM_HIP(hipMemcpyAsync(p->dNd, normDirect, p->NP * 16, hipMemcpyHostToDevice, p->stream));
    if (O) FSGM_HIP(hipMemcpyAsync(p->dO, O, p->NP * 8, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

// test.m:45-50 on the uploaded map: dIn -> dOut (filterD1), dD2 (filterD2), dDisp
static fsgm_status post_enqueue(fsgm_post_plan* p, double vMax, double n, double dMax) {
    PostGeom g{p->dPd0, p->dNd, p->dO, vMax, n};
    const int W = p->W, H = p->H;
    launch_speckle_filter(p->stream, p->dIn, p->dA, nullptr, p->dParent, p->dSize, p->dScan, W, H, 2.0, 100.0);      // :45
    launch_disp_from_first(p->stream, p->dA, p->dD2, g, W, H);                                                        // :46
    launch_fb_check(p->stream, p->dA, p->dD2, p->dB, g, W, H);                                                        // :47
    launch_speckle_filter(p->stream, p->dB, p->dA, nullptr, p->dParent, p->dSize, p->dScan, W, H, dMax,
                          (double)H * (double)W / 10.0);                                                              // :48 rows*cols/10
    launch_scanline_in_fill(p->stream, p->dA, p->dOut, p->dLeft, W, H);                                               // :49
    launch_vzind2disp(p->stream, p->dOut, p->dO, p->dDisp, p->NP, vMax, n);                                           // :50
    FSGM_HIP(hipGetLastError());
    return FSGM_OK;
}

fsgm_status fsgm_post_plan_run(fsgm_post_plan* p, double vMax, double n, double dMax) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_HIP(hipSetDevice(p->device));
    return post_enqueue(p, vMax, n, dMax);
}

fsgm_status fsgm_post_plan_download(fsgm_post_plan* p, double* filterD1, double* filterD2, double* disp) {
    FSGM_REQUIRE(p, "null plan");
    FSGM_HIP(hipSetDevice(p->device));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    if (filterD1) FSGM_HIP(hipMemcpy(filterD1, p->dOut, p->NP * 8, hipMemcpyDeviceToHost));
    if (filterD2) FSGM_HIP(hipMemcpy(filterD2, p->dD2, p->NP * 8, hipMemcpyDeviceToHost));
    if (disp) FSGM_HIP(hipMemcpy(disp, p->dDisp, p->NP * 8, hipMemcpyDeviceToHost));
    return FSGM_OK;
}

fsgm_status fsgm_post_plan_time(fsgm_post_plan* p, double vMax, double n, double dMax, int32_t warmup, int32_t iters, float* ms_avg) {
    FSGM_REQUIRE(p && ms_avg && iters >= 1 && warmup >= 0, "fsgm_post_plan_time: bad argument");
    FSGM_HIP(hipSetDevice(p->device));
    fsgm_status st;
    for (int i = 0; i < warmup; i++)
        if ((st = post_enqueue(p, vMax, n, dMax)) != FSGM_OK) return st;
    FSGM_HIP(hipEventRecord(p->ev0, p->stream));
    for (int i = 0; i < iters; i++)
        if ((st = post_enqueue(p, vMax, n, dMax)) != FSGM_OK) return st;
    FSGM_HIP(hipEventRecord(p->ev1, p->stream));
    FSGM_HIP(hipEventSynchronize(p->ev1));
    float ms = 0;
    FSGM_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *ms_avg = ms / iters;
    return FSGM_OK;
}

// ---- one host-pointer entry point per MATLAB function, on a cached plan ----
static std::mutex g_post_mu;
static std::vector<fsgm_post_plan*> g_post_cache;

void fsgm_post_shutdown_internal(void) {
    std::lock_guard<std::mutex> lk(g_post_mu);
    for (fsgm_post_plan* p : g_post_cache) fsgm_post_plan_destroy(p);
    g_post_cache.clear();
}

static fsgm_status cached_plan(fsgm_post_plan** out, int W, int H, int device) {
    for (fsgm_post_plan* q : g_post_cache)
        if (q->W == W && q->H == H && q->device == device) { *out = q; return hipSetDevice(device) == hipSuccess ? FSGM_OK : fail(FSGM_ERR_HIP, "hipSetDevice failed"); }
    fsgm_status st = fsgm_post_plan_create(out, W, H, device);
    if (st != FSGM_OK) return st;
    if (g_post_cache.size() >= 2) {
        fsgm_post_plan_destroy(g_post_cache.front());
        g_post_cache.erase(g_post_cache.begin());
    }
    g_post_cache.push_back(*out);
    return FSGM_OK;
}

static fsgm_status require_non_negative(const double* D1, size_t n, const char* fn) {
    for (size_t i = 0; i < n; i++)
        if (D1[i] < 0.0) return fail(FSGM_ERR_INVALID, "%s: D1 must hold non-negative values or NaN (element %zu is %g)", fn, i, D1[i]);
    return FSGM_OK;
}

fsgm_status fsgm_speckle_filter_host(const double* image, int32_t W, int32_t H, double maxDiff, double maxSpeckleSize,
                                     double* imageFiltered, int32_t* labelImage, int32_t device) {
    FSGM_REQUIRE(image && imageFiltered, "fsgm_speckle_filter: null argument");
    std::lock_guard<std::mutex> lk(g_post_mu);
    fsgm_post_plan* p;
    fsgm_status st = cached_plan(&p, W, H, device);
    if (st != FSGM_OK) return st;
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    FSGM_HIP(hipMemcpyAsync(p->dIn, image, p->NP * 8, hipMemcpyHostToDevice, p->stream));
    launch_speckle_filter(p->stream, p->dIn, p->dOut, labelImage ? p->dLabels : nullptr, p->dParent, p->dSize, p->dScan,
                          W, H, maxDiff, maxSpeckleSize);
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipMemcpyAsync(imageFiltered, p->dOut, p->NP * 8, hipMemcpyDeviceToHost, p->stream));
    if (labelImage) FSGM_HIP(hipMemcpyAsync(labelImage, p->dLabels, p->NP * 4, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_calc_disp_from_first_host(const double* D1, int32_t W, int32_t H, const double* Pd0, const double* normDirect,
                                           const double* O, double vMax, double n, double* D2, int32_t device) {
    FSGM_REQUIRE(D1 && Pd0 && normDirect && O && D2, "fsgm_calc_disp_from_first: null argument");
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1");
    fsgm_status st = require_non_negative(D1, (size_t)W * H, "fsgm_calc_disp_from_first");
    if (st != FSGM_OK) return st;
    std::lock_guard<std::mutex> lk(g_post_mu);
    fsgm_post_plan* p;
    if ((st = cached_plan(&p, W, H, device)) != FSGM_OK) return st;
    if ((st = fsgm_post_plan_upload(p, D1, Pd0, normDirect, O)) != FSGM_OK) return st;
    launch_disp_from_first(p->stream, p->dIn, p->dD2, PostGeom{p->dPd0, p->dNd, p->dO, vMax, n}, W, H);
    FSGM_HIP(hipGetLastError());
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    FSGM_HIP(hipMemcpyAsync(D2, p->dD2, p->NP * 8, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_forward_backward_check_host(const double* D1, const double* D2, int32_t W, int32_t H, const double* Pd0,
                                             const double* normDirect, const double* O, double vMax, double n,
                                             double* D1checked, int32_t device) {
    FSGM_REQUIRE(D1 && D2 && Pd0 && normDirect && O && D1checked, "fsgm_forward_backward_check: null argument");
    std::lock_guard<std::mutex> lk(g_post_mu);
    fsgm_post_plan* p;
    fsgm_status st = cached_plan(&p, W, H, device);
    if (st != FSGM_OK) return st;
    if ((st = fsgm_post_plan_upload(p, D1, Pd0, normDirect, O)) != FSGM_OK) return st;
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    FSGM_HIP(hipMemcpyAsync(p->dD2, D2, p->NP * 8, hipMemcpyHostToDevice, p->stream));
    launch_fb_check(p->stream, p->dIn, p->dD2, p->dOut, PostGeom{p->dPd0, p->dNd, p->dO, vMax, n}, W, H);
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipMemcpyAsync(D1checked, p->dOut, p->NP * 8, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_scanline_in_fill_host(const double* input, int32_t W, int32_t H, double* output, int32_t device) {
    FSGM_REQUIRE(input && output, "fsgm_scanline_in_fill: null argument");
    std::lock_guard<std::mutex> lk(g_post_mu);
    fsgm_post_plan* p;
    fsgm_status st = cached_plan(&p, W, H, device);
    if (st != FSGM_OK) return st;
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    FSGM_HIP(hipMemcpyAsync(p->dIn, input, p->NP * 8, hipMemcpyHostToDevice, p->stream));
    launch_scanline_in_fill(p->stream, p->dIn, p->dOut, p->dLeft, W, H);
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipMemcpyAsync(output, p->dOut, p->NP * 8, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_vzind2disp_host(const double* w, const double* O, int32_t W, int32_t H, double vMax, double n, double* D, int32_t device) {
    FSGM_REQUIRE(w && O && D, "fsgm_vzind2disp: null argument");
    std::lock_guard<std::mutex> lk(g_post_mu);
    fsgm_post_plan* p;
    fsgm_status st = cached_plan(&p, W, H, device);
    if (st != FSGM_OK) return st;
    StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
    FSGM_HIP(hipMemcpyAsync(p->dIn, w, p->NP * 8, hipMemcpyHostToDevice, p->stream));
    FSGM_HIP(hipMemcpyAsync(p->dO, O, p->NP * 8, hipMemcpyHostToDevice, p->stream));
    launch_vzind2disp(p->stream, p->dIn, p->dO, p->dDisp, p->NP, vMax, n);
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipMemcpyAsync(D, p->dDisp, p->NP * 8, hipMemcpyDeviceToHost, p->stream));
    FSGM_HIP(hipStreamSynchronize(p->stream));
    guard.dismiss();
    return FSGM_OK;
}

fsgm_status fsgm_vmf_host(const double* flow, int32_t W, int32_t H, int32_t channels, double* flowMed, int32_t device) {
    FSGM_REQUIRE(flow && flowMed, "fsgm_vmf: null argument");
    FSGM_REQUIRE(channels >= 1 && channels <= 3, "fsgm_vmf: 1..3 channels (got %d)", channels);
    std::lock_guard<std::mutex> lk(g_post_mu);
    fsgm_post_plan* p;
    fsgm_status st = cached_plan(&p, W, H, device);
    if (st != FSGM_OK) return st;
    double* src[3] = {p->dIn, p->dA, p->dB};                     // one plane per scratch map
    double* dst[3] = {p->dOut, p->dD2, p->dDisp};
    for (int c = 0; c < channels; c++) {
        StreamGuard guard(p->stream);   // an early exit drains the stream: queued copies use the caller's memory
        FSGM_HIP(hipMemcpyAsync(src[c], flow + (size_t)c * p->NP, p->NP * 8, hipMemcpyHostToDevice, p->stream));
        launch_vmf(p->stream, src[c], dst[c], W, H, 1);
        FSGM_HIP(hipMemcpyAsync(flowMed + (size_t)c * p->NP, dst[c], p->NP * 8, hipMemcpyDeviceToHost, p->stream));
    }
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipStreamSynchronize(p->stream));
    return FSGM_OK;
}

fsgm_status fsgm_epi_postprocess_host(const double* D1, int32_t W, int32_t H, const double* Pd0, const double* normDirect,
                                      const double* O, double vMax, double n, double dMax,
                                      double* filterD1, double* filterD2, double* disp, int32_t device) {
    FSGM_REQUIRE(D1 && Pd0 && normDirect && O && filterD1, "fsgm_epi_postprocess: null argument");
    FSGM_REQUIRE(W >= 1 && H >= 1, "width/height must be >= 1");
    fsgm_status st = require_non_negative(D1, (size_t)W * H, "fsgm_epi_postprocess");
    if (st != FSGM_OK) return st;
    std::lock_guard<std::mutex> lk(g_post_mu);
    fsgm_post_plan* p;
    if ((st = cached_plan(&p, W, H, device)) != FSGM_OK) return st;
    if ((st = fsgm_post_plan_upload(p, D1, Pd0, normDirect, O)) != FSGM_OK) return st;
    if ((st = post_enqueue(p, vMax, n, dMax)) != FSGM_OK) return st;
    return fsgm_post_plan_download(p, filterD1, filterD2, disp);
}

}  // extern "C"
