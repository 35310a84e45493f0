// fsgm_device.h -- shared device-side helpers for the gfx950 kernels.
// wave = 64 lanes; all lane-group tricks below assume it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsgm {

// ---- x86-64 gcc double->integer conversions (cvttsd2si).  The reference is C++ built for
// x86-64; out-of-range and NaN inputs give the "integer indefinite" there, whereas the gfx950
// v_cvt_* instructions saturate.  Bit-exact parity needs the former.
__device__ __forceinline__ int32_t f64_to_i32_x86(double v) {
    return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;
}
__device__ __forceinline__ int64_t f64_to_i64_x86(double v) {
    return (v >= -9223372036854775808.0 && v < 9223372036854775808.0) ? (int64_t)v : INT64_MIN;
}
__device__ __forceinline__ uint32_t f64_to_u32_x86(double v) { return (uint32_t)(uint64_t)f64_to_i64_x86(v); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(hi, max(lo, v)); }

// (int)round(v) exactly as x86-64 computes it -- C round() is half away from zero, the cast is
// cvttsd2si (INT_MIN for NaN and anything outside int) -- in 8 instructions:
//   t = trunc(v), f = v - t (exact, |f| < 1); the increment is trunc(2f) = -1, 0 or +1, i.e. |f| >= 0.5
//   with v's sign; v_cvt_i32_f64 of v itself is t as an integer.  A rounded value of exactly 2^31
//   comes out of the wrapping integer add as INT_MIN as well; |v| >= 2^31, inf and NaN fail the
//   one range compare (the conversions' results are not used then).
__device__ __forceinline__ int32_t round_to_i32_x86(double v) {
    const double t = trunc(v);
    const double f = __dsub_rn(v, t);
    const uint32_t n = (uint32_t)__double2int_rz(v);
    const uint32_t inc = (uint32_t)__double2int_rz(__dadd_rn(f, f));
    return fabs(v) < 2147483648.0 ? (int32_t)(n + inc) : INT32_MIN;
}

// clamp to [0, hi] in one instruction (hi >= 0)
__device__ __forceinline__ int clamp0(int v, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(hi));
    return r;
}

// clamp((int)round(v), 0, hi) for a finite |v| < 2^30 (the caller has proved it), three full-rate instructions:
//   trunc(fl(v + pred(0.5))) is C round() -- half away from zero -- for every 0 <= v < 2^51: with h = pred(0.5) =
//   0.5 - 2^-54 the exact sum v + h lies 2^-54 below v + 0.5, and (i) v = k + 0.5: the sum k + 1 - 2^-54 is nearer to
//   k + 1 than to the double below it (a tie for k = 0, which goes to the even mantissa of 1.0), so it rounds UP to
//   k + 1; (ii) v below k + 0.5 by at least its own ulp: the sum stays below k + 1 after rounding, because the
//   doubles below k + 1 are no coarser than ulp(v) >= 2^-53 > 2^-54 apart from the sum; (iii) v < 0.5: sum < 1.
//   (Adding 0.5 itself fails at v = pred(0.5): 1 - 2^-54 ties up to 1.0.)  Negative v: v > -0.5 gives a sum in
//   (-2^-54, 0.5) -> 0; v <= -0.5 gives a sum <= 0 that truncates to <= 0 and the clamp returns 0, as it does for
//   the reference's round(v) <= -1.  tests/test_capi_cpu.py checks the identity around every half integer.
__device__ __forceinline__ int round_clamp_small(double v, int hi) {
    return clamp0(__double2int_rz(__dadd_rn(v, 0x1.fffffffffffffp-2)), hi);
}

// parabola vertex offset as the pyramidal variant writes it (calc_pyd_cost_sgm.cpp:341-344)
__device__ __forceinline__ double pyd_parabola(double cl, double c0, double cr) {
    return cr < cl ? __ddiv_rn(__ddiv_rn(__dsub_rn(cr, cl), __dsub_rn(c0, cl)), 2.0)
                   : __ddiv_rn(__ddiv_rn(__dsub_rn(cr, cl), __dsub_rn(c0, cr)), 2.0);
}

// ---- packed 2 x u16 arithmetic (VOP3P v_pk_*_u16) ----
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
// {hi:lo} >> 16, low 32 bits  ==  (lo >> 16) | (hi << 16)      (v_alignbit_b32)
__device__ __forceinline__ uint32_t align16(uint32_t hi, uint32_t lo) {
    return __builtin_amdgcn_alignbit(hi, lo, 16);
}

// 16-byte streaming store / load (written once, read once by a later kernel: keep it out of the
// way of lines that are re-read)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(void* p, const uint4 v) {
    u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (u32x4*)p);
}
__device__ __forceinline__ uint4 load_nt(const void* p) {
    const u32x4 t = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(t.x, t.y, t.z, t.w);
}

// ---- DPP controls (gfx9 encoding) ----
constexpr int DPP_QUAD_1032 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_2301 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_SHL1 = 0x101;        // lane i <- lane i+1 (within a row of 16)
constexpr int DPP_ROW_SHR1 = 0x111;        // lane i <- lane i-1
constexpr int DPP_WAVE_SHL1 = 0x130;       // lane i <- lane i+1 across the whole wave
constexpr int DPP_WAVE_SHR1 = 0x138;       // lane i <- lane i-1 across the whole wave
constexpr int DPP_ROW_MIRROR = 0x140;      // i <-> 15-i
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // i <-> 7-i within each half row

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}

// min over a group of G adjacent lanes (G = 1,2,4,8,16,32, group aligned), result in every lane.
// v_min_u32 with the DPP modifier on its first source: one instruction per butterfly level
// (the compiler emits mov + mov_dpp + min for the builtin form).  The s_nop covers the
// VALU-write -> DPP-read hazard (2 wait states), which hipcc does not pad inside asm.
#define FSGM_MIN_DPP(x, ctrl) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(x))
template <int G>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t x) {
    if (G >= 2)  FSGM_MIN_DPP(x, "quad_perm:[1,0,3,2]");
    if (G >= 4)  FSGM_MIN_DPP(x, "quad_perm:[2,3,0,1]");
    if (G >= 8)  FSGM_MIN_DPP(x, "row_half_mirror");
    if (G >= 16) FSGM_MIN_DPP(x, "row_mirror");
    if (G >= 32) {                                            // rows 0<->1 and 2<->3 (gfx950 v_permlane16_swap)
        uint32_t y = x;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
        x = min(x, y);
    }
    return x;
}

// minimum over the 64 lanes of a wave, uniform result: DPP within the rows of 16, then the four row results
// through scalar registers (all lanes must be active)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
    x = group_min_u32<16>(x);
    const uint32_t r0 = __builtin_amdgcn_readlane(x, 0), r1 = __builtin_amdgcn_readlane(x, 16);
    const uint32_t r2 = __builtin_amdgcn_readlane(x, 32), r3 = __builtin_amdgcn_readlane(x, 48);
    return min(min(r0, r1), min(r2, r3));
}

}  // namespace fsgm
