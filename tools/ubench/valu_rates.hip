// valu_rates.hip -- issue rate of the fp64 VALU instructions the epipolar cost kernel is made of, on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/ubench/valu_rates tools/ubench/valu_rates.hip
// Prints cycles per wave64 instruction per SIMD at the clock the card reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITER = 2048, CH = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed, int iters) {
    double a[CH];
    int32_t n[CH];
    float g[CH];
    for (int i = 0; i < CH; i++) { a[i] = seed + threadIdx.x * 1e-3 + i; n[i] = threadIdx.x + i; g[i] = (float)a[i]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 3) asm volatile("v_trunc_f64 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(a[i]));
            if (OP == 5) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(seed) : "vcc");
            if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(it));
            if (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(g[i]) : "v"(g[(i + 1) % CH]));
            if (OP == 8) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(n[i]) : "v"(it));
            if (OP == 9) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(n[i]) : "v"(it));
            if (OP == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(n[i]) : "v"(it));
            if (OP == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(it));
        }
    }
    double s = 0;
    for (int i = 0; i < CH; i++) s += a[i] + n[i] + g[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
void run(const char* name, double* d, double mhz) {
    const int blocks = 256 * 8;                           // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITER * CH;           // wave instructions
    const double cyc = ms * 1e-3 * mhz * 1e6 * 1024.0 / winstr;     // 1024 SIMDs
    printf("%-16s %8.3f ms  %6.2f cycles per wave instruction per SIMD\n", name, ms, cyc);
}

int main() {
    double* d; hipMalloc(&d, 8);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const double mhz = pr.clockRate / 1000.0;
    printf("%s, %d CUs, %.0f MHz\n", pr.name, pr.multiProcessorCount, mhz);
    run<6>("v_add_u32", d, mhz);  run<7>("v_fma_f32", d, mhz);  run<8>("v_med3_i32", d, mhz);  run<9>("v_bcnt_u32_b32", d, mhz);
    run<10>("v_mad_u32_u24", d, mhz); run<11>("v_mul_lo_u32", d, mhz);
    run<0>("v_add_f64", d, mhz);  run<1>("v_mul_f64", d, mhz);  run<2>("v_fma_f64", d, mhz);
    run<3>("v_trunc_f64", d, mhz); run<4>("v_cvt_i32_f64", d, mhz); run<5>("v_cmp_lt_f64", d, mhz);
    return 0;
}
