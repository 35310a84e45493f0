// counter_calib.hip -- kernels of KNOWN behaviour for reading the SQ counters of tools/sq_counters.sh against:
//   calib_valu_pk      8 waves per SIMD issuing nothing but independent v_pk_maximum3_f16: the VALU issue ceiling for the
//                      packed instructions the SGM step is made of ("VALU busy" of a saturated SIMD)
//   calib_valu_vop2    the same with v_and_b32 (32-bit encoding)
//   calib_valu_f64     the same with v_mul_f64, calib_valu_cvt64 with v_cvt_i32_f64 (the cost fill's instruction classes)
//   calib_lds_b128     every lane reads and writes 16 contiguous bytes (lane * 16): the conflict-free ds_read_b128 /
//                      ds_write_b128 pattern of the sweeps' diagonal states -- what SQ_LDS_BANK_CONFLICT reads for it
//   calib_lds_conf32   ds_write_b32 with a 128-byte lane stride: every lane of a group on one bank (32-way conflict)
//   calib_lds_row8     ds_write_b32 x 8 at dword offsets j*8 + g*64 + i: the WTA rows' pattern of wta_row_record
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/counter_calib tools/ubench/counter_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void calib_valu_pk(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8], b = seed + threadIdx.x;
    for (int i = 0; i < 8; i++) a[i] = (seed * (i + 3) + threadIdx.x) & 0x03FF03FFu;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_valu_vop2(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8], b = seed + threadIdx.x;
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 0x12345678u) out[0] = s;
}
// the cost fill's instruction classes: fp64 multiply / add and the fp64 -> integer conversion
__global__ __launch_bounds__(256) void calib_valu_f64(double* out, double seed, int iters) {
    double a[8], b = seed + threadIdx.x * 1e-3;
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 0.12345) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_valu_cvt64(uint32_t* out, double seed, int iters) {
    double a[8];
    uint32_t r[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (i + 3) + threadIdx.x; r[i] = 0; }
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r[i]) : "v"(a[i]));
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += r[i];
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_lds_b128(uint32_t* out, int iters) {
    __shared__ uint4 buf[2][256];
    const int t = threadIdx.x;
    uint4 v = make_uint4(t, t + 1, t + 2, t + 3);
    buf[0][t] = v; buf[1][t] = v;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        const uint4 r = buf[it & 1][t];
        v.x += r.x; v.y ^= r.y; v.z += r.z; v.w ^= r.w;
        buf[(it & 1) ^ 1][t] = v;
        __syncthreads();
    }
    if (v.x == 0x12345678u) out[0] = v.y;
}
__global__ __launch_bounds__(256) void calib_lds_conf32(uint32_t* out, int iters) {
    __shared__ uint32_t buf[256 * 32];
    const int t = threadIdx.x;
    uint32_t v = t;
    for (int it = 0; it < iters; it++) {
        buf[t * 32] = v;                                   // 128-byte stride: one bank
        __syncthreads();
        v += buf[((t + 1) & 255) * 32];
        __syncthreads();
    }
    if (v == 0x12345678u) out[0] = v;
}
__global__ __launch_bounds__(256) void calib_lds_row8(uint32_t* out, int iters) {
    __shared__ uint32_t buf[256 * 8];
    const int t = threadIdx.x, j = t & 7, g = t >> 3;
    uint32_t v = t;
    volatile uint32_t* row = buf + g * 64 + j * 8;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) row[i] = v + i;
        __syncthreads();
        v += buf[(t * 9) & 2047];
        __syncthreads();
    }
    if (v == 0x12345678u) out[0] = v;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4096);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_valu_pk, dim3(256 * 8), dim3(256), 0, 0, d, 12345u, 4096);
        hipLaunchKernelGGL(calib_valu_vop2, dim3(256 * 8), dim3(256), 0, 0, d, 12345u, 4096);
        hipLaunchKernelGGL(calib_valu_f64, dim3(256 * 8), dim3(256), 0, 0, (double*)d, 1.000001, 2048);
        hipLaunchKernelGGL(calib_valu_cvt64, dim3(256 * 8), dim3(256), 0, 0, d, 1.5, 2048);
        hipLaunchKernelGGL(calib_lds_b128, dim3(256 * 4), dim3(256), 0, 0, d, 4096);
        hipLaunchKernelGGL(calib_lds_conf32, dim3(256 * 2), dim3(256), 0, 0, d, 1024);
        hipLaunchKernelGGL(calib_lds_row8, dim3(256 * 4), dim3(256), 0, 0, d, 2048);
    }
    hipDeviceSynchronize();
    printf("counter_calib done\n");
    return 0;
}
