// capi_common.h -- error plumbing shared by the C-ABI translation units
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/fsgm.h"

namespace fsgm {

char* last_error_buf();   // thread-local, 512 bytes

inline fsgm_status fail(fsgm_status st, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return st;
}

#define FSGM_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::fsgm::fail(FSGM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define FSGM_REQUIRE(cond, ...)                                          \
    do {                                                                 \
        if (!(cond)) return ::fsgm::fail(FSGM_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// The host-pointer entry points keep their cached plans / arenas PER DEVICE and hold that device's lock for the length of a
// call: calls on different devices run side by side (fsgm_*_batch_devices_host starts one host thread per device of its
// list), calls on one device take turns -- they would share the GPU anyway.
constexpr int FSGM_MAX_DEVICES = 64;
template <class T>
struct PerDevice {
    std::mutex mu[FSGM_MAX_DEVICES];
    T v[FSGM_MAX_DEVICES];
};
#define FSGM_DEVICE_SLOT(dev) FSGM_REQUIRE((dev) >= 0 && (dev) < ::fsgm::FSGM_MAX_DEVICES, "device %d out of range", (int)(dev))

// Scope guard for host-pointer entry points: work queued on `st` may still read the caller's input
// buffers or write its output buffers (async copies), so every exit that is not the normal one
// (which has synchronised already and calls dismiss()) drains the stream before the caller gets
// control back.
struct StreamGuard {
    hipStream_t st;
    bool armed = true;
    explicit StreamGuard(hipStream_t s) : st(s) {}
    void dismiss() { armed = false; }
    ~StreamGuard() { if (armed) (void)hipStreamSynchronize(st); }
    StreamGuard(const StreamGuard&) = delete;
    StreamGuard& operator=(const StreamGuard&) = delete;
};

}  // namespace fsgm
