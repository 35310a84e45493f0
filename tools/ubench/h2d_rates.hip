// h2d_rates.hip -- how to move one calc_cost_sgm call's inputs (19.5 MB) and outputs (3.7 MB) across PCIe:
// (A) hipMemcpyAsync from / to pageable memory, (B) pin the caller's buffers in place for the call
// (hipHostRegister / Unregister), (C) copy into a resident pinned staging buffer with N threads, then DMA.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/h2d_rates tools/ubench/h2d_rates.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t IN = 19500000, OUT = 3726000;
    char* h = (char*)malloc(IN); char* ho = (char*)malloc(OUT);
    memset(h, 1, IN); memset(ho, 0, OUT);
    char *d, *dout, *pin, *pino;
    (void)hipMalloc(&d, IN); (void)hipMalloc(&dout, OUT);
    (void)hipHostMalloc(&pin, IN); (void)hipHostMalloc(&pino, OUT);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto timeit = [&](const char* name, auto fn) {
        fn(); fn();
        const int it = 20;
        const double t0 = now_ms();
        for (int i = 0; i < it; i++) fn();
        const double ms = (now_ms() - t0) / it;
        printf("%-58s %7.3f ms  (%5.1f GB/s on %zu bytes)\n", name, ms, (IN + OUT) / ms / 1e6, IN + OUT);
    };
    timeit("A  pageable hipMemcpyAsync H2D + D2H, one sync", [&]() {
        (void)hipMemcpyAsync(d, h, IN, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(ho, dout, OUT, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
    });
    timeit("A' pageable hipMemcpy H2D (5 pieces) + D2H (2 pieces)", [&]() {
        for (int k = 0; k < 5; k++) (void)hipMemcpyAsync(d + k * (IN / 5), h + k * (IN / 5), IN / 5, hipMemcpyHostToDevice, s);
        for (int k = 0; k < 2; k++) (void)hipMemcpyAsync(ho + k * (OUT / 2), dout + k * (OUT / 2), OUT / 2, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
    });
    timeit("B  hipHostRegister in place + DMA + unregister", [&]() {
        (void)hipHostRegister(h, IN, hipHostRegisterDefault); (void)hipHostRegister(ho, OUT, hipHostRegisterDefault);
        (void)hipMemcpyAsync(d, h, IN, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(ho, dout, OUT, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipHostUnregister(h); (void)hipHostUnregister(ho);
    });
    timeit("P  already pinned buffers (lower bound)", [&]() {
        (void)hipMemcpyAsync(d, pin, IN, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(pino, dout, OUT, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
    });
    for (int nt : {1, 2, 4, 8}) {
        char name[96]; snprintf(name, sizeof name, "C  memcpy -> pinned staging (%d threads, 8 chunks) + DMA", nt);
        timeit(name, [&]() {
            const int NCH = 8; const size_t ch = IN / NCH;
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back([&, t]() { for (int c = t; c < NCH; c += nt) memcpy(pin + c * ch, h + c * ch, ch); });
            for (auto& x : th) x.join();
            (void)hipMemcpyAsync(d, pin, IN, hipMemcpyHostToDevice, s);
            (void)hipMemcpyAsync(pino, dout, OUT, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            memcpy(ho, pino, OUT);
        });
    }
    timeit("C' single thread: memcpy chunk k+1 while chunk k is in flight", [&]() {
        const int NCH = 8; const size_t ch = IN / NCH;
        for (int c = 0; c < NCH; c++) { memcpy(pin + c * ch, h + c * ch, ch); (void)hipMemcpyAsync(d + c * ch, pin + c * ch, ch, hipMemcpyHostToDevice, s); }
        (void)hipMemcpyAsync(pino, dout, OUT, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        memcpy(ho, pino, OUT);
    });
    return 0;
}
