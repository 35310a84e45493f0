// capi_multi.hip -- a batch of frames over several GPUs of one node FROM ONE PROCESS (include/fsgm.h, "device lists").
//
// The path shards by frame: frame i goes to entry i mod n of the caller's device list, every entry runs its
// frames through the single-device batch call on a host thread of its own, nothing is exchanged between devices -- no
// collective, no peer copy; RCCL has no part in it.  That is what a MATLAB session (one process, host code unchanged) needs
// to use the 8 GPUs of a node: the Python-side sharding (fsgm_amd/batch.py, bench.py under torchrun) is one process per GPU.
// List entries are HIP ordinals taken modulo the number of devices present, so the list {0, 1, 2, 3} runs on a one-GPU box
// too (four slots sharing the device: the per-device lock of the host entry points serialises them), and results never
// depend on the list.
#include "capi_common.h"
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

using namespace fsgm;

namespace {

struct SlotResult {
    fsgm_status st = FSGM_OK;
    std::string msg;
};

// devices[i] modulo the device count (negative entries are an error)
fsgm_status resolve_devices(int32_t n_devices, const int32_t* devices, std::vector<int>& out) {
    FSGM_REQUIRE(n_devices >= 1 && devices, "device list: empty");
    FSGM_REQUIRE(n_devices <= 1024, "device list: %d entries", n_devices);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FSGM_ERR_HIP, "no HIP device available (libfsgm_hip has no CPU fallback)");
    out.resize(n_devices);
    for (int i = 0; i < n_devices; i++) {
        FSGM_REQUIRE(devices[i] >= 0, "device list: entry %d is negative (%d)", i, devices[i]);
        out[i] = devices[i] % ndev;
    }
    return FSGM_OK;
}

// Runs call(slot, device, frame indices of the slot) for every slot that owns a frame, each on its own thread (the first
// slot on the calling thread), and returns the first failure in slot order with its message.
template <class Call>
fsgm_status run_slots(int n_frames, const std::vector<int>& dev, Call call) {
    const int S = (int)dev.size();
    std::vector<std::vector<int>> mine(S);
    for (int i = 0; i < n_frames; i++) mine[i % S].push_back(i);
    std::vector<SlotResult> res(S);
    auto work = [&](int s) {
        res[s].st = call(s, dev[s], mine[s]);
        if (res[s].st != FSGM_OK) res[s].msg = last_error_buf();      // the worker's thread-local message
    };
    std::vector<std::thread> th;
    for (int s = 1; s < S; s++)
        if (!mine[s].empty()) th.emplace_back(work, s);
    if (!mine[0].empty()) work(0);
    for (std::thread& t : th) t.join();
    for (int s = 0; s < S; s++)
        if (res[s].st != FSGM_OK) return fail(res[s].st, "device list entry %d (device %d): %s", s, dev[s], res[s].msg.c_str());
    return FSGM_OK;
}

template <class T>
std::vector<T> pick(const T* a, const std::vector<int>& idx) {
    std::vector<T> v;
    v.reserve(idx.size());
    for (int i : idx) v.push_back(a[i]);
    return v;
}

}  // namespace

extern "C" {

int32_t fsgm_parse_device_list(const char* text, int32_t* devices, int32_t max_devices) {
    if (!text || !devices || max_devices <= 0) return 0;
    int32_t n = 0;
    const char* p = text;
    while (*p && n < max_devices) {
        while (*p == ' ' || *p == ',' || *p == ';') p++;
        if (!*p) break;
        char* end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || v > 1 << 20) return -1;
        devices[n++] = (int32_t)v;
        p = end;
    }
    return n;
}

void fsgm_shard_frames(int32_t n_frames, int32_t n_devices, int32_t slot, int32_t* frames, int32_t* count) {
    int32_t c = 0;
    if (n_devices > 0 && slot >= 0 && slot < n_devices)
        for (int32_t i = slot; i < n_frames; i += n_devices) { if (frames) frames[c] = i; c++; }
    if (count) *count = c;
}

fsgm_status fsgm_calc_cost_sgm_batch_devices_host(int32_t n, const fsgm_epi_in* in, const fsgm_epi_out* out, const fsgm_epi_params* prm,
                                                  int32_t n_devices, const int32_t* devices) {
    FSGM_REQUIRE(n >= 1 && in && out, "fsgm_calc_cost_sgm: null argument");
    std::vector<int> dev;
    fsgm_status st = resolve_devices(n_devices, devices, dev);
    if (st != FSGM_OK) return st;
    const fsgm_epi_params base = prm ? *prm : fsgm_epi_params_default();
    return run_slots(n, dev, [&](int, int device, const std::vector<int>& idx) {
        fsgm_epi_params pr = base;
        pr.device = device;
        const std::vector<fsgm_epi_in> i = pick(in, idx);
        const std::vector<fsgm_epi_out> o = pick(out, idx);
        return fsgm_calc_cost_sgm_batch_host((int32_t)idx.size(), i.data(), o.data(), &pr);
    });
}

fsgm_status fsgm_calc_pyd_cost_sgm_batch_devices_host(int32_t n, const fsgm_pyd_in* in, const fsgm_pyd_out* out,
                                                      int32_t n_devices, const int32_t* devices) {
    FSGM_REQUIRE(n >= 1 && in && out, "fsgm_calc_pyd_cost_sgm: null argument");
    std::vector<int> dev;
    fsgm_status st = resolve_devices(n_devices, devices, dev);
    if (st != FSGM_OK) return st;
    return run_slots(n, dev, [&](int, int device, const std::vector<int>& idx) {
        const std::vector<fsgm_pyd_in> i = pick(in, idx);
        const std::vector<fsgm_pyd_out> o = pick(out, idx);
        return fsgm_calc_pyd_cost_sgm_batch_host((int32_t)idx.size(), i.data(), o.data(), device);
    });
}

fsgm_status fsgm_calc_pyd_cost_sgm_ng_batch_devices_host(int32_t n, const fsgm_ng_in* in, const fsgm_ng_out* out,
                                                         int32_t n_devices, const int32_t* devices) {
    FSGM_REQUIRE(n >= 1 && in && out, "fsgm_calc_pyd_cost_sgm_ng: null argument");
    std::vector<int> dev;
    fsgm_status st = resolve_devices(n_devices, devices, dev);
    if (st != FSGM_OK) return st;
    return run_slots(n, dev, [&](int, int device, const std::vector<int>& idx) {
        const std::vector<fsgm_ng_in> i = pick(in, idx);
        const std::vector<fsgm_ng_out> o = pick(out, idx);
        return fsgm_calc_pyd_cost_sgm_ng_batch_host((int32_t)idx.size(), i.data(), o.data(), device);
    });
}

fsgm_status fsgm_calc_cost_sgm_ng_batch_devices_host(int32_t n, const fsgm_otf_in* in, const fsgm_otf_out* out,
                                                     int32_t n_devices, const int32_t* devices) {
    FSGM_REQUIRE(n >= 1 && in && out, "fsgm_calc_cost_sgm_ng: null argument");
    std::vector<int> dev;
    fsgm_status st = resolve_devices(n_devices, devices, dev);
    if (st != FSGM_OK) return st;
    // The reference draws from libc rand() in raster order, frame after frame (calc_cost_sgm_ng.cpp:148-149; process-global
    // state).  Frames without a stream of their own get their draws here, on the calling thread, in frame order -- what a
    // sequence of single calls would have drawn -- before the frames scatter over the devices.
    std::vector<fsgm_otf_in> frames(in, in + n);
    std::vector<std::vector<int32_t>> drawn(n);
    for (int i = 0; i < n; i++)
        if (!frames[i].rand_stream) {
            FSGM_REQUIRE(frames[i].width >= 1 && frames[i].height >= 1, "bad image size");
            drawn[i].resize((size_t)fsgm_sgm_ng_rand_draws(frames[i].width, frames[i].height));
            for (int32_t& v : drawn[i]) v = rand();
            frames[i].rand_stream = drawn[i].data();
        }
    return run_slots(n, dev, [&](int, int device, const std::vector<int>& idx) {
        const std::vector<fsgm_otf_in> i = pick(frames.data(), idx);
        const std::vector<fsgm_otf_out> o = pick(out, idx);
        return fsgm_calc_cost_sgm_ng_batch_host((int32_t)idx.size(), i.data(), o.data(), device);
    });
}

fsgm_status fsgm_pyramidal_sgm_batch_devices_host(int32_t n, const fsgm_pyramid_pair* pairs, int32_t width, int32_t height, int32_t channels,
                                                  const fsgm_pyramid_params* prm, int32_t n_devices, const int32_t* devices) {
    FSGM_REQUIRE(n >= 1 && pairs && prm, "fsgm_pyramidal_sgm: null argument");
    std::vector<int> dev;
    fsgm_status st = resolve_devices(n_devices, devices, dev);
    if (st != FSGM_OK) return st;
    return run_slots(n, dev, [&](int, int device, const std::vector<int>& idx) {
        fsgm_pyramid_params pr = *prm;
        pr.device = device;
        for (int i : idx) {
            const fsgm_status s = fsgm_pyramidal_sgm_host(pairs[i].I0, pairs[i].I1, width, height, channels, &pr, pairs[i].mv, pairs[i].minC, pairs[i].mvPyd);
            if (s != FSGM_OK) return s;
        }
        return FSGM_OK;
    });
}

fsgm_status fsgm_pyramidal_sgm_ng_batch_devices_host(int32_t n, const fsgm_pyramid_pair* pairs, int32_t width, int32_t height, int32_t channels,
                                                     const fsgm_ng_pyramid_params* prm, int32_t n_devices, const int32_t* devices) {
    FSGM_REQUIRE(n >= 1 && pairs && prm, "fsgm_pyramidal_sgm_ng: null argument");
    std::vector<int> dev;
    fsgm_status st = resolve_devices(n_devices, devices, dev);
    if (st != FSGM_OK) return st;
    return run_slots(n, dev, [&](int, int device, const std::vector<int>& idx) {
        fsgm_ng_pyramid_params pr = *prm;
        pr.device = device;
        for (int i : idx) {
            const fsgm_status s = fsgm_pyramidal_sgm_ng_host(pairs[i].I0, pairs[i].I1, width, height, channels, &pr, pairs[i].mv, pairs[i].minC, pairs[i].mvPyd);
            if (s != FSGM_OK) return s;
        }
        return FSGM_OK;
    });
}

}  // extern "C"
