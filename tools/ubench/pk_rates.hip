// pk_rates.hip -- issue rates of the packed-u16 / cross-lane instructions the SGM path step is made of (gfx950), at
// 1, 2, 4 and 8 waves per SIMD, with 8 independent chains per wave and with one dependent chain; plus a
// bit-exactness probe of v_pk_minimum3_f16 used as an unsigned 16-bit 3-input minimum (valid for patterns < 0x7C00
// when fp16 denormals are not flushed).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/pk_rates tools/ubench/pk_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int ITER = 16384;

#define OPS(X) \
    X(0, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1") \
    X(1, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1") \
    X(2, "v_pk_minimum3_f16", "v_pk_minimum3_f16 %0, %0, %1, %1") \
    X(3, "v_pk_min_f16", "v_pk_min_f16 %0, %0, %1") \
    X(4, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1") \
    X(5, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 16") \
    X(6, "v_mov_b32_dpp row_shr:1", "s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(7, "v_min_u32_dpp quad_perm", "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(8, "v_min_u16_sdwa", "v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0") \
    X(9, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc") \
    X(10, "v_pk_sub_u16 op_sel_hi", "v_pk_sub_u16 %0, %0, %1 op_sel_hi:[1,0]") \
    X(11, "v_min3_u32", "v_min3_u32 %0, %0, %1, %1") \
    X(12, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %1") \
    X(13, "v_add_u32", "v_add_u32 %0, %0, %1") \
    X(14, "v_pk_max_u16", "v_pk_max_u16 %0, %0, %1") \
    X(15, "v_pk_min_i16", "v_pk_min_i16 %0, %0, %1") \
    X(16, "v_permlane32_swap", "s_nop 1\n\tv_permlane32_swap_b32 %0, %1") \
    X(17, "v_mov_b32_dpp row_shr:8", "s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf") \
    X(18, "v_min_u16 (VOP2)", "v_min_u16 %0, %0, %1") \
    X(19, "v_min_u32 (VOP2)", "v_min_u32 %0, %0, %1") \
    X(20, "v_and_b32 (VOP2)", "v_and_b32 %0, %0, %1") \
    X(21, "v_mov_b32 (VOP1)", "v_mov_b32 %0, %1") \
    X(22, "v_cndmask_b32 e32 vcc", "v_cndmask_b32 %0, %0, %1, vcc") \
    X(23, "v_cndmask_b32 e64 sgpr", "v_cndmask_b32 %0, %0, %1, s[20:21]") \
    X(24, "v_mov_dpp no nop", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(25, "v_min_u32_dpp src1 no nop", "v_min_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(26, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 16, %1") \
    X(27, "pk_min + add_u32 pair", "v_pk_min_u16 %0, %0, %1\n\tv_add_u32 %0, %0, %1") \
    X(28, "v_pk_add_u16 inline const", "v_pk_add_u16 %0, %0, 6 op_sel_hi:[1,0]") \
    X(29, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %1") \
    X(30, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1") \
    X(31, "v_sub_u32 (VOP2)", "v_sub_u32 %0, %0, %1") \
    X(32, "v_pk_maximum3_f16", "v_pk_maximum3_f16 %0, %0, %1, %1") \
    X(33, "v_pk_sub_u16 clamp", "v_pk_sub_u16 %0, %0, %1 clamp") \
    X(34, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %0, 8, %0") \
    X(35, "v_or_b32 (VOP2)", "v_or_b32 %0, %0, %1") \
    X(36, "v_lshrrev_b32 (VOP2)", "v_lshrrev_b32 %0, 8, %0")

template <int OP, int CH>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[CH];
    uint32_t b = seed + threadIdx.x;
    for (int i = 0; i < CH; i++) a[i] = seed * (i + 3) + threadIdx.x;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_mov_b64 s[20:21], vcc" : : "v"(b), "v"(seed) : "vcc", "s20", "s21");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8 / CH; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
#define X(N, NAME, ASM) if (OP == N) asm volatile(ASM : "+v"(a[i]) : "v"(b));
                OPS(X)
#undef X
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < CH; i++) s += a[i];
    if (s == 0x12345678u) out[0] = s;
}

// ds_bpermute: LDS crossbar, not a VALU instruction
template <int CH>
__global__ __launch_bounds__(256) void kbperm(uint32_t* out, uint32_t seed, int iters) {
    int a[CH];
    const int addr = ((threadIdx.x + 8) & 63) * 4;
    for (int i = 0; i < CH; i++) a[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8 / CH; r++)
#pragma unroll
            for (int i = 0; i < CH; i++) a[i] = __builtin_amdgcn_ds_bpermute(addr, a[i]);
    }
    int s = 0;
    for (int i = 0; i < CH; i++) s += a[i];
    if (s == 0x12345678) out[0] = s;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(16);
    hipEventRecord(e0);
    launch(ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

template <int OP>
void run(const char* name, uint32_t* d, double mhz) {
    printf("%-26s", name);
    for (int wps : {1, 2, 4, 8}) {                          // waves per SIMD = 256-thread blocks per CU
        const int blocks = 256 * wps;
        const double ms = time_ms([&](int it) { hipLaunchKernelGGL((k<OP, 8>), dim3(blocks), dim3(256), 0, 0, d, 12345u, it); });
        const double winstr = (double)blocks * 4 * ITER * 8;
        printf("  %dw %5.2f", wps, ms * 1e-3 * mhz * 1e6 * 1024.0 / winstr);
    }
    {                                                       // one dependent chain, 1 wave per SIMD: latency
        const double ms = time_ms([&](int it) { hipLaunchKernelGGL((k<OP, 1>), dim3(256), dim3(256), 0, 0, d, 12345u, it); });
        printf("  dep-chain %5.2f", ms * 1e-3 * mhz * 1e6 * 1024.0 / ((double)256 * 4 * ITER * 8));
    }
    printf("   (cycles per wave instruction per SIMD)\n");
}

__global__ void probe_min3(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* o, uint32_t* o2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r, r2;
    asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r2) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    o[i] = r; o2[i] = r2;
}

int main() {
    uint32_t* d; hipMalloc(&d, 64);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const double mhz = pr.clockRate / 1000.0;
    printf("%s, %d CUs, %.0f MHz; 8 independent chains per wave unless noted\n", pr.name, pr.multiProcessorCount, mhz);
#define X(N, NAME, ASM) run<N>(NAME, d, mhz);
    OPS(X)
#undef X
    printf("%-26s", "ds_bpermute_b32");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        const double ms = time_ms([&](int it) { hipLaunchKernelGGL((kbperm<8>), dim3(blocks), dim3(256), 0, 0, d, 12345u, it); });
        printf("  %dw %5.2f", wps, ms * 1e-3 * mhz * 1e6 * 1024.0 / ((double)blocks * 4 * ITER * 8));
    }
    printf("   (cycles per wave instruction per SIMD-equivalent)\n");

    // bit-exactness of the f16 minimum as an integer minimum
    const int n = 1 << 20;
    std::vector<uint32_t> ha(n), hb(n), hc(n), ho(n), ho2(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 20); };
    auto val = [&](int i) -> uint32_t {                      // a u16 below 0x7C00, biased towards the small (denormal-pattern) range
        const uint32_t r = rnd();
        switch (i & 3) { case 0: return r % 0x400; case 1: return r % 0x7C00; case 2: return r % 300; default: return (r % 2) ? 0 : r % 0x7C00; }
    };
    for (int i = 0; i < n; i++) {
        ha[i] = val(i) | (val(i + 1) << 16); hb[i] = val(i + 2) | (val(i + 3) << 16); hc[i] = val(i + 1) | (val(i + 2) << 16);
    }
    uint32_t *da, *db, *dc, *dout, *dout2;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dout2, n * 4);
    hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_min3, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, dout2, n);
    hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ho2.data(), dout2, n * 4, hipMemcpyDeviceToHost);
    long bad3 = 0, bad2 = 0;
    for (int i = 0; i < n; i++) {
        auto m = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
        const uint32_t lo = m(m(ha[i] & 0xFFFF, hb[i] & 0xFFFF), hc[i] & 0xFFFF), hi = m(m(ha[i] >> 16, hb[i] >> 16), hc[i] >> 16);
        if (ho[i] != (lo | (hi << 16))) { if (bad3 < 5) printf("min3 mismatch %08x %08x %08x -> %08x want %08x\n", ha[i], hb[i], hc[i], ho[i], lo | (hi << 16)); bad3++; }
        auto M = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
        const uint32_t lo2 = M(M(ha[i] & 0xFFFF, hb[i] & 0xFFFF), hc[i] & 0xFFFF), hi2 = M(M(ha[i] >> 16, hb[i] >> 16), hc[i] >> 16);
        if (ho2[i] != (lo2 | (hi2 << 16))) { if (bad2 < 5) printf("max3 mismatch %08x %08x %08x -> %08x want %08x\n", ha[i], hb[i], hc[i], ho2[i], lo2 | (hi2 << 16)); bad2++; }
    }
    printf("v_pk_minimum3_f16 as u16 min3 on %d random triples below 0x7C00: %ld mismatches; v_pk_maximum3_f16 as u16 max3: %ld mismatches\n", n, bad3, bad2);
    return 0;
}
