// pattern_rates.hip -- what HBM delivers for the aggregation kernels' access patterns with the arithmetic taken out:
//   sweep: workgroups of 256 lanes own a strip of 32 columns (4 KB per image row), 16 rows per launch, 24 launches per
//          pass over H = 375 rows; per row one 16-byte load from each of two volumes and one 16-byte store (the final
//          sweep's C + Y_dn + Y_h in, nothing out is cheaper; the down sweep's C in, Y_dn out);
//   pair:  a wave owns 8 image rows and walks along x, 128 bytes per row and step (16 bytes a lane), 4 steps requested
//          ahead; one volume in, one out (the sum pass).
// Frames x 1242 x 375 x 128 bytes per volume, as in bench.py.  GB/s = bytes read + written / time.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/pattern_rates tools/ubench/pattern_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int W = 1242, H = 375, D = 128;

template <int NIN, bool OUT, int COLS = 32>
__global__ __launch_bounds__(256) void sweep_pat(const uint4* __restrict__ A, const uint4* __restrict__ B, uint4* __restrict__ O, int y0, int rows, uint32_t* sink) {
    const int x0 = blockIdx.x * COLS + (threadIdx.x >> 3);
    const int j = threadIdx.x & 7;
    const size_t f = blockIdx.y, vol = (size_t)W * H * 8;      // uint4 units
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < rows; r++)
      for (int xs = 0; xs < COLS; xs += 32) {
        const int x = x0 + xs;
        if (x >= W) continue;
        const size_t idx = f * vol + ((size_t)(y0 + r) * W + x) * 8 + j;
        uint4 v = A[idx];
        if (NIN > 1) { const uint4 w = B[idx]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        if (OUT) O[idx] = v; else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
      }
    if (!OUT && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

__global__ __launch_bounds__(256) void pair_pat(const uint4* __restrict__ A, uint4* __restrict__ O) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = (blockIdx.x * 4 + wave) * 8 + (lane >> 3), j = lane & 7;
    if ((blockIdx.x * 4 + wave) * 8 >= H) return;
    const int rc = row < H ? row : H - 1;
    const size_t f = blockIdx.y, base = f * (size_t)W * H * 8 + (size_t)rc * W * 8 + j;
    uint4 ring[4];
#pragma unroll
    for (int i = 0; i < 4; i++) ring[i] = A[base + (size_t)i * 8];
    for (int t0 = 0; t0 < W; t0 += 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int t = t0 + i;
            if (t >= W) break;
            const uint4 v = ring[i];
            ring[i] = A[base + (size_t)(t + 4 < W ? t + 4 : W - 1) * 8];
            O[base + (size_t)t * 8] = v;
        }
    }
}

template <class F>
static double ms_of(F run) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run();
    (void)hipEventRecord(e0);
    const int it = 5;
    for (int i = 0; i < it; i++) run();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / it;
}

int main() {
    const int F = 40;
    const size_t vol = (size_t)W * H * D * F;
    uint4 *a, *b, *o; uint32_t* sink;
    (void)hipMalloc(&a, vol); (void)hipMalloc(&b, vol); (void)hipMalloc(&o, vol); (void)hipMalloc(&sink, 4);
    (void)hipMemset(a, 1, vol); (void)hipMemset(b, 2, vol);
    hipStream_t s1, s2;
    (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    const dim3 g((W + 31) / 32, F);
    auto pass = [&](auto kern, hipStream_t st, int f0, int nf) {
        for (int y0 = 0; y0 < H; y0 += 16) {
            const int rows = H - y0 < 16 ? H - y0 : 16;
            hipLaunchKernelGGL(kern, dim3((W + 31) / 32, nf), dim3(256), 0, st, a + (size_t)f0 * W * H * 8, b + (size_t)f0 * W * H * 8, o + (size_t)f0 * W * H * 8, y0, rows, sink);
        }
    };
    printf("%d frames of %dx%dx%d, GB/s read+written\n", F, W, H, D);
    double ms = ms_of([&]() { pass(sweep_pat<1, true>, 0, 0, F); });
    printf("sweep pattern, 1 in 1 out, one stream          %7.0f  (%.3f ms)\n", 2.0 * vol / (ms * 1e-3) / 1e9, ms);
    ms = ms_of([&]() { pass(sweep_pat<2, false>, 0, 0, F); });
    printf("sweep pattern, 2 in 0 out, one stream          %7.0f  (%.3f ms)\n", 2.0 * vol / (ms * 1e-3) / 1e9, ms);
    ms = ms_of([&]() { pass(sweep_pat<2, true>, 0, 0, F); });
    printf("sweep pattern, 2 in 1 out, one stream          %7.0f  (%.3f ms)\n", 3.0 * vol / (ms * 1e-3) / 1e9, ms);
    ms = ms_of([&]() { pass(sweep_pat<2, true>, s1, 0, F / 2); pass(sweep_pat<2, true>, s2, F / 2, F / 2); (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2); });
    printf("sweep pattern, 2 in 1 out, two lanes of frames %7.0f  (%.3f ms)\n", 3.0 * vol / (ms * 1e-3) / 1e9, ms);
    {
        auto pass64 = [&](hipStream_t st, int f0, int nf) {
            for (int y0 = 0; y0 < H; y0 += 16) {
                const int rows = H - y0 < 16 ? H - y0 : 16;
                hipLaunchKernelGGL((sweep_pat<2, true, 64>), dim3((W + 63) / 64, nf), dim3(256), 0, st, a + (size_t)f0 * W * H * 8, b + (size_t)f0 * W * H * 8, o + (size_t)f0 * W * H * 8, y0, rows, sink);
            }
        };
        ms = ms_of([&]() { pass64(0, 0, F); });
        printf("sweep pattern, 64-column strips, 2 in 1 out     %7.0f  (%.3f ms)\n", 3.0 * vol / (ms * 1e-3) / 1e9, ms);
        ms = ms_of([&]() { pass64(s1, 0, F / 2); pass64(s2, F / 2, F / 2); (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2); });
        printf("sweep pattern, 64-column strips, two lanes      %7.0f  (%.3f ms)\n", 3.0 * vol / (ms * 1e-3) / 1e9, ms);
    }
    ms = ms_of([&]() { hipLaunchKernelGGL(pair_pat, dim3((H + 31) / 32, F), dim3(256), 0, 0, a, o); });
    printf("pair pattern, 1 in 1 out                       %7.0f  (%.3f ms)\n", 2.0 * vol / (ms * 1e-3) / 1e9, ms);
    ms = ms_of([&]() { hipLaunchKernelGGL(pair_pat, dim3((H + 31) / 32, F), dim3(256), 0, s1, a, o); pass(sweep_pat<2, true>, s2, 0, F); (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2); });
    printf("pair 1+1 beside sweep 2+1 (5 volumes moved)    %7.0f  (%.3f ms)\n", 5.0 * vol / (ms * 1e-3) / 1e9, ms);
    return 0;
}
