// capi_ng.hip -- C ABI for the two neighbour-guided variants (include/fsgm.h).
// (SURVEY 8(a) a-11..a-14).  Device memory comes from a process-lifetime arena (NgPool below).
#include "capi_common.h"
#include "epi_kernels.h"
#include "ng_kernels.h"
#include <algorithm>
#include <mutex>
#include <stdlib.h>
#include <utility>
#include <vector>

using namespace fsgm;

namespace {
// Device memory of the two variants comes from one arena that lives as long as the process (like the plans
// behind the other host entry points: a MATLAB session calls the MEX once per frame, and allocating and
// freeing 0.6 GB per call costs more than the kernels).  One call at a time, as in a MEX.
struct NgPool {
    std::mutex mu;
    int device = -1;
    char* base = nullptr;
    size_t cap = 0;
    int small_calls = 0;               // consecutive calls that used at most a quarter of the arena
    hipStream_t stream = nullptr;
};
NgPool g_pools[FSGM_MAX_DEVICES];      // one arena per device: calls on different devices run side by side

struct DevBufs {                       // the arena of `device` for the duration of one call
    NgPool& g_pool;
    std::unique_lock<std::mutex> lk;
    explicit DevBufs(int device) : g_pool(g_pools[device]), lk(g_pools[device].mu) {}
    std::vector<std::pair<void**, size_t>> req;
    hipStream_t stream = nullptr;
    // Every exit drains the arena's stream before the pool mutex (declared first, released last) lets the next
    // call re-carve the arena or the caller frees its buffers: an early error return may leave copies from the
    // caller's memory and kernels queued.  On the normal path the stream is already idle.
    ~DevBufs() { if (stream) (void)hipStreamSynchronize(stream); }
    void want(void** p, size_t bytes) { req.emplace_back(p, (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255); }
    hipError_t commit(int device) {
        size_t total = 0;
        for (auto& r : req) total += r.second;
        hipError_t e = hipSuccess;
        // the arena grows to the largest call and shrinks again once eight calls in a row used at most a quarter of it
        // (a session that moved on to smaller frames does not hold the large frames' memory until fsgm_shutdown)
        g_pool.small_calls = (g_pool.cap >= (1u << 24) && total <= g_pool.cap / 4) ? g_pool.small_calls + 1 : 0;
        if (g_pool.device != device || g_pool.cap < total || g_pool.small_calls >= 8) {
            g_pool.small_calls = 0;
            if (g_pool.base) { (void)hipSetDevice(g_pool.device); (void)hipFree(g_pool.base); (void)hipSetDevice(device); }
            if (g_pool.stream && g_pool.device != device) { (void)hipStreamDestroy(g_pool.stream); g_pool.stream = nullptr; }
            g_pool.base = nullptr; g_pool.cap = 0; g_pool.device = device;
            if ((e = hipMalloc((void**)&g_pool.base, total)) != hipSuccess) return e;
            g_pool.cap = total;
        }
        if (!g_pool.stream && (e = hipStreamCreateWithFlags(&g_pool.stream, hipStreamNonBlocking)) != hipSuccess) return e;
        stream = g_pool.stream;
        size_t used = 0;
        for (auto& r : req) { *r.first = g_pool.base + used; used += r.second; }
        return hipSuccess;
    }
};

fsgm_status pick_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FSGM_ERR_HIP, "no HIP device available (libfsgm_hip has no CPU fallback)");
    FSGM_REQUIRE(device >= 0 && device < ndev && device < FSGM_MAX_DEVICES, "device %d out of range (have %d)", device, ndev);
    FSGM_HIP(hipSetDevice(device));
    return FSGM_OK;
}
}  // namespace

extern "C" {

void fsgm_ng_shutdown_internal(void) {
    for (NgPool& g_pool : g_pools) {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        if (g_pool.device >= 0) (void)hipSetDevice(g_pool.device);
        if (g_pool.base) (void)hipFree(g_pool.base);
        if (g_pool.stream) (void)hipStreamDestroy(g_pool.stream);
        g_pool.base = nullptr; g_pool.cap = 0; g_pool.stream = nullptr; g_pool.device = -1;
    }
}

fsgm_status fsgm_calc_pyd_cost_sgm_ng_batch_host(int32_t n, const fsgm_ng_in* in, const fsgm_ng_out* out, int32_t device) {
    FSGM_REQUIRE(n >= 1 && in && out, "fsgm_calc_pyd_cost_sgm_ng: null argument");
    const fsgm_ng_in& a = in[0];
    for (int i = 0; i < n; i++) {
        FSGM_REQUIRE(in[i].I1 && in[i].I2 && in[i].preMv && out[i].minC && out[i].flow,
                     "fsgm_calc_pyd_cost_sgm_ng: frame %d has a null pointer", i);
        const fsgm_ng_in& b = in[i];
        FSGM_REQUIRE(b.width == a.width && b.height == a.height && b.mvWidth == a.mvWidth && b.mvHeight == a.mvHeight &&
                     b.halfSearchWinSize == a.halfSearchWinSize && b.aggSize == a.aggSize &&
                     b.subPixelRefine == a.subPixelRefine && b.P1 == a.P1 && b.P2 == a.P2,
                     "frames of one batch must share shape and parameters (frame %d differs)", i);
    }
    const int W = a.width, H = a.height, r = a.halfSearchWinSize, rAgg = a.aggSize / 2;
    FSGM_REQUIRE(W >= 1 && H >= 1 && a.mvWidth >= 1 && a.mvHeight >= 1, "bad image / hint-map size");
    FSGM_REQUIRE(r >= 0 && a.aggSize >= 0, "halfSearchWinSize and aggSize must be >= 0");
    const long long D = 9LL * (2 * r + 1) * (2 * r + 1);
    if (D > FSGM_NG_MAX_D) return fail(FSGM_ERR_UNSUPPORTED, "%lld candidates per pixel exceed %d", D, FSGM_NG_MAX_D);
    if ((double)W * H * D >= 2147483648.0) return fail(FSGM_ERR_UNSUPPORTED, "candidate volume exceeds 2^31 entries per frame");
    fsgm_status st = pick_device(device);
    if (st != FSGM_OK) return st;
    const size_t NP = (size_t)W * H, MV = (size_t)a.mvWidth * a.mvHeight, N = NP * D, B = n;
    DevBufs d(device);
    uint8_t *dI1, *dI2, *dDk; uint16_t *dDd, *dCm; uint32_t* dCk; uint32_t *dCen1, *dCen2, *dS, *dMinC, *dUnsafe, *dBox, *dKstat; double *dMv, *dFlow; Cand* dC; int16_t* dL4;
    d.want((void**)&dUnsafe, 8);                       // { a vector beyond the packed matcher's range, a key that does not fit 4 bytes }
    d.want((void**)&dI1, B * NP);
    d.want((void**)&dI2, B * NP);
    d.want((void**)&dCen1, B * NP * 4);
    d.want((void**)&dCen2, B * NP * 4);
    d.want((void**)&dMv, B * MV * 16);
    d.want((void**)&dC, B * N * sizeof(Cand));
    d.want((void**)&dS, B * N * 4);
    d.want((void**)&dMinC, B * NP * 4);
    d.want((void**)&dFlow, B * NP * 16);
    d.want((void**)&dDd, B * N * 2);
    d.want((void**)&dDk, B * NP);
    d.want((void**)&dBox, B * NP * 4);
    d.want((void**)&dKstat, NG_KSTAT_WORDS * 4);
    d.want((void**)&dCk, B * N * 4);
    d.want((void**)&dCm, B * N * 2);
    d.want((void**)&dL4, D <= 128 ? B * NP * NG_L4_PER_PIXEL * sizeof(int16_t) : 0);
    { const hipError_t e = d.commit(device); if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? FSGM_ERR_NOMEM : FSGM_ERR_HIP, "fsgm_calc_pyd_cost_sgm_ng: %s", hipGetErrorString(e)); }
    for (int i = 0; i < n; i++) {
        FSGM_HIP(hipMemcpyAsync(dI1 + i * NP, in[i].I1, NP, hipMemcpyHostToDevice, d.stream));
        FSGM_HIP(hipMemcpyAsync(dI2 + i * NP, in[i].I2, NP, hipMemcpyHostToDevice, d.stream));
        FSGM_HIP(hipMemcpyAsync(dMv + i * 2 * MV, in[i].preMv, MV * 16, hipMemcpyHostToDevice, d.stream));
    }
    // 4-byte candidate entries (the 3x3 hint kernel's sizes, D = 81): the keys live in S's memory until the matchers need S
    static const bool k4_env = [] { const char* e = getenv("FSGM_NG_K4"); return !(e && e[0] == '0'); }();      // A/B switch
    static const bool hint_env = [] { const char* e = getenv("FSGM_NG_COST_HINT"); return !(e && e[0] == '0'); }();
    const bool dedupe_on = [] { const char* e = getenv("FSGM_NG_DEDUPE"); return !(e && atoi(e) == 0); }();
    const bool k4 = k4_env && hint_env && dedupe_on && r == 1 && rAgg == 1 && D <= 128;
    if (!k4) FSGM_HIP(hipMemsetAsync(dS, 0, B * N * 4, d.stream));              // :111
    FSGM_HIP(hipMemsetAsync(dUnsafe, 0, 8, d.stream));
    launch_census(d.stream, dI1, dCen1, W, H, n);                               // :485-486
    launch_census(d.stream, dI2, dCen2, W, H, n);
    NgCostArgs ca;
    ca.K4 = k4 ? dS : nullptr; ca.flags = k4 ? dUnsafe : nullptr;
    ca.cen1 = dCen1; ca.cen2 = dCen2; ca.mv = dMv; ca.C = dC; ca.unsafe = dUnsafe; ca.W = W; ca.H = H;
    ca.mvW = a.mvWidth; ca.mvH = a.mvHeight; ca.rAgg = rAgg; ca.rX = r; ca.rY = r;
    launch_ng_cost(d.stream, ca, n);
    NgAggArgs ga;
    ga.C = dC; ga.S = dS; ga.unsafe = dUnsafe; ga.W = W; ga.H = H; ga.D = (int)D; ga.P1 = a.P1; ga.P2 = a.P2;
    ga.dd = nullptr; ga.dk = nullptr; ga.dbox = nullptr; ga.kstat = nullptr; ga.ck = nullptr; ga.cm = nullptr; ga.L4 = nullptr;
    if (D <= 128) {                                  // repeats in the candidate lists: the matchers scan each distinct entry once
        launch_ng_dedupe(d.stream, dC, dDd, dDk, dBox, dKstat, dCk, dCm, W, H, (int)D, n, ca.K4, ca.flags);
        ga.dd = dDd; ga.dk = dDk; ga.dbox = dBox; ga.kstat = dKstat; ga.ck = dCk; ga.cm = dCm; ga.L4 = dL4;
        if (k4) launch_ng_prepare_matchers(d.stream, ga, dS, dC, dUnsafe, n);
    }
    launch_ng_aggregate(d.stream, ga, n);
    NgWtaArgs wa;
    wa.C = dC; wa.S = dS; wa.minC = dMinC; wa.flow = dFlow; wa.W = W; wa.H = H; wa.D = (int)D;
    wa.cm = ga.dd ? dCm : nullptr; wa.dk = ga.dd ? dDk : nullptr; wa.L4 = ga.L4; wa.kstat = ga.kstat; wa.K4 = ca.K4; wa.flags = ca.flags;
    launch_ng_wta(d.stream, wa, n);
    bool want_S = false;
    for (int i = 0; i < n; i++) want_S = want_S || out[i].S;
    if (want_S && ga.L4) launch_ng_l4_to_s(d.stream, dS, dL4, dCm, dDk, dKstat, W, H, (int)D, n);   // the compact kernel leaves its sums in L4
    if (want_S && ga.dd) launch_ng_fill_repeats(d.stream, dS, dDd, dCm, W, H, (int)D, n);      // the compact kernel adds to kept entries only
    if (a.subPixelRefine) {                                                     // :516-517
        NgSubpixArgs sa;
        sa.cen1 = dCen1; sa.cen2 = dCen2; sa.flow = dFlow; sa.W = W; sa.H = H;
        launch_ng_subpixel(d.stream, sa, n);
    }
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipStreamSynchronize(d.stream));
    for (int i = 0; i < n; i++) {
        FSGM_HIP(hipMemcpy(out[i].minC, dMinC + i * NP, NP * 4, hipMemcpyDeviceToHost));
        FSGM_HIP(hipMemcpy(out[i].flow, dFlow + i * 2 * NP, NP * 16, hipMemcpyDeviceToHost));
        if (out[i].S) FSGM_HIP(hipMemcpy(out[i].S, dS + i * N, N * 4, hipMemcpyDeviceToHost));
    }
    return FSGM_OK;
}

fsgm_status fsgm_calc_pyd_cost_sgm_ng_host(const fsgm_ng_in* in, const fsgm_ng_out* out, int32_t device) {
    return fsgm_calc_pyd_cost_sgm_ng_batch_host(1, in, out, device);
}

int64_t fsgm_sgm_ng_rand_draws(int32_t W, int32_t H) { return (int64_t)W * H * 8; }

fsgm_status fsgm_calc_cost_sgm_ng_batch_host(int32_t n, const fsgm_otf_in* in, const fsgm_otf_out* out, int32_t device) {
    FSGM_REQUIRE(n >= 1 && in && out, "fsgm_calc_cost_sgm_ng: null argument");
    const int W = in[0].width, H = in[0].height;
    FSGM_REQUIRE(W >= 1 && H >= 1, "bad image size");
    for (int i = 0; i < n; i++) {
        FSGM_REQUIRE(in[i].I1 && in[i].I2 && out[i].minC && out[i].flow, "fsgm_calc_cost_sgm_ng: frame %d has a null pointer", i);
        FSGM_REQUIRE(in[i].width == W && in[i].height == H && in[i].P1 == in[0].P1 && in[i].P2 == in[0].P2,
                     "frames of one batch must share shape and parameters (frame %d differs)", i);
    }
    fsgm_status st = pick_device(device);
    if (st != FSGM_OK) return st;
    const size_t NP = (size_t)W * H, B = n, rowE = (size_t)W * OTF_E;
    DevBufs d(device);
    uint8_t *dI1, *dI2; uint32_t *dCen1, *dCen2, *dMinC; double* dFlow; int32_t* dRnd; Cand* dLrow;
    d.want((void**)&dI1, B * NP);
    d.want((void**)&dI2, B * NP);
    d.want((void**)&dCen1, B * NP * 4);
    d.want((void**)&dCen2, B * NP * 4);
    d.want((void**)&dRnd, B * NP * 8 * 4);
    d.want((void**)&dLrow, B * 6 * rowE * sizeof(Cand));
    d.want((void**)&dMinC, B * NP * 4);
    d.want((void**)&dFlow, B * NP * 16);
    { const hipError_t e = d.commit(device); if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? FSGM_ERR_NOMEM : FSGM_ERR_HIP, "fsgm_calc_cost_sgm_ng: %s", hipGetErrorString(e)); }
    std::vector<int32_t> drawn;
    for (int i = 0; i < n; i++) {
        const int32_t* rs = in[i].rand_stream;
        if (!rs) {                                   // the reference's own source of hints: libc rand()
            drawn.resize(NP * 8);
            for (size_t k = 0; k < NP * 8; k++) drawn[k] = rand();
            rs = drawn.data();
        }
        FSGM_HIP(hipMemcpy(dRnd + i * NP * 8, rs, NP * 8 * 4, hipMemcpyHostToDevice));
        FSGM_HIP(hipMemcpyAsync(dI1 + i * NP, in[i].I1, NP, hipMemcpyHostToDevice, d.stream));
        FSGM_HIP(hipMemcpyAsync(dI2 + i * NP, in[i].I2, NP, hipMemcpyHostToDevice, d.stream));
    }
    FSGM_HIP(hipMemsetAsync(dLrow, 0, B * 6 * rowE * sizeof(Cand), d.stream));  // :205-207
    launch_census(d.stream, dI1, dCen1, W, H, n);                               // :233-234
    launch_census(d.stream, dI2, dCen2, W, H, n);
    OtfArgs oa;
    oa.I1 = dI1; oa.cen1 = dCen1; oa.cen2 = dCen2; oa.rnd = dRnd; oa.Lrow = dLrow; oa.minC = dMinC; oa.flow = dFlow;
    oa.W = W; oa.H = H; oa.P1 = in[0].P1; oa.P2 = in[0].P2;
    const char* ex = getenv("FSGM_OTF_EXACT");
    oa.exact = (ex && atoi(ex) != 0) ? 1 : 0;
    launch_otf(d.stream, oa, n);
    FSGM_HIP(hipGetLastError());
    FSGM_HIP(hipStreamSynchronize(d.stream));
    for (int i = 0; i < n; i++) {
        FSGM_HIP(hipMemcpy(out[i].minC, dMinC + i * NP, NP * 4, hipMemcpyDeviceToHost));
        FSGM_HIP(hipMemcpy(out[i].flow, dFlow + i * 2 * NP, NP * 16, hipMemcpyDeviceToHost));
    }
    return FSGM_OK;
}

fsgm_status fsgm_calc_cost_sgm_ng_host(const fsgm_otf_in* in, const fsgm_otf_out* out, int32_t device) {
    return fsgm_calc_cost_sgm_ng_batch_host(1, in, out, device);
}

}  // extern "C"
