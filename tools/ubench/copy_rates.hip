// copy_rates.hip -- what a plain HIP copy kernel reaches on this device, by shape (blocks, bytes in flight per lane,
// temporal hint): the ceiling the aggregation's HBM traffic is priced against.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/copy_rates tools/ubench/copy_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copyk(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16) {
    // each block owns a contiguous chunk; U loads in flight per lane
    const size_t per_block = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block < n16 ? b0 + per_block : n16;
    for (size_t i = b0 + threadIdx.x; i < b1; i += (size_t)256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + (size_t)u * 256 < b1) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + (size_t)u * 256 < b1) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
    }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copyflat(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16) {
    // one pass: block b handles elements [b*256*U, (b+1)*256*U)
    const size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i + (size_t)u * 256 < n16) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; u++) if (i + (size_t)u * 256 < n16) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
}

template <class F>
static double gbps(F launch, size_t bytes) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    (void)hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; i++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * bytes * it / (ms * 1e-3) / 1e9;
}

int main() {
    const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
    u32x4 *a, *b;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 1, bytes);
    printf("copy of %zu MiB, GB/s read+written\n", bytes >> 20);
    printf("hipMemcpyAsync D2D            %7.0f\n", gbps([&]() { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes));
#define CHUNK(U, NT, BL) printf("chunked U=%d nt=%d blocks=%-6d %7.0f\n", U, NT, BL, gbps([&]() { hipLaunchKernelGGL((copyk<U, NT>), dim3(BL), dim3(256), 0, 0, b, a, n16); }, bytes));
    CHUNK(4, false, 1024) CHUNK(4, false, 2048) CHUNK(4, false, 4096) CHUNK(4, false, 8192) CHUNK(8, false, 2048) CHUNK(8, false, 4096)
    CHUNK(2, false, 4096) CHUNK(4, true, 2048) CHUNK(4, true, 4096) CHUNK(8, true, 2048) CHUNK(1, false, 8192) CHUNK(1, false, 16384)
#define FLAT(U, NT) printf("flat    U=%d nt=%d              %7.0f\n", U, NT, gbps([&]() { hipLaunchKernelGGL((copyflat<U, NT>), dim3((unsigned)((n16 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, b, a, n16); }, bytes));
    FLAT(1, false) FLAT(2, false) FLAT(4, false) FLAT(8, false) FLAT(1, true) FLAT(4, true)
    return 0;
}
