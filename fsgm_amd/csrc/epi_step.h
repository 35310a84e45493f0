// epi_step.h -- the DP step of calc_cost_sgm.cpp:33-66 in the mirrored variable, the register / byte layouts around it and the
// per-pixel WTA record, shared by the fused aggregation kernels (epi_sweep.hip: block sweeps, pair kernels;
// epi_band.hip: band sweeps).  Derivation and layout: the header of epi_sweep.hip.
#pragma once
#include "fsgm_device.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsgm {

namespace {

constexpr uint32_t SEL_ODD = 0x0C030C01u;    // v_perm: bytes 1,3 -> 2 x u16
constexpr uint32_t SEL_PACK = 0x06020400u;   // v_perm(b, a): bytes a.0, b.0, a.2, b.2
constexpr uint32_t SEL_NB = 0x05040302u;     // v_perm(a, b): (b.hi16, a.lo16)
constexpr uint32_t SEL_NB_NOLO = 0x05040C0Cu;   // ... with 0 in the low half  (no d-1 neighbour)
constexpr uint32_t SEL_NB_NOHI = 0x0C0C0302u;   // ... with 0 in the high half (no d+1 neighbour)

__device__ __forceinline__ uint32_t pk_mad16(uint32_t a, uint32_t b, uint32_t c) {        // v_pk_mad_u16
    u16x2 r = __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c);
    return __builtin_bit_cast(uint32_t, r);
}
// max(a - b, 0) per half (v_pk_sub_u16 clamp)
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
// 3-input maximum / minimum of packed u16 values below 0x7C00 (see the header: fp16 order = integer order there);
// hipcc fuses the nested 2-input forms into v_pk_maximum3_f16 / v_pk_minimum3_f16 on gfx950
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
    f16x2 r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
                                            __builtin_bit_cast(f16x2, c));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c) {
    f16x2 r = __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
                                            __builtin_bit_cast(f16x2, c));
    return __builtin_bit_cast(uint32_t, r);
}
// min(x.lo, x.hi) in the low half, zero above (v_min_u16 with SDWA half selects)
__device__ __forceinline__ uint32_t min_halves(uint32_t x) {
    uint32_t r;
    asm("v_min_u16_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(r) : "v"(x));
    return r;
}

// 16 natural-order cost bytes of a lane -> CP[i] = (C[i] + P2, C[i+8] + P2)
__device__ __forceinline__ void unpack_c(const uint4 w, uint32_t (&CP)[8], const uint32_t P2pk) {
    CP[0] = pk_add(__builtin_amdgcn_perm(w.z, w.x, 0x0C040C00u), P2pk);
    CP[1] = pk_add(__builtin_amdgcn_perm(w.z, w.x, 0x0C050C01u), P2pk);
    CP[2] = pk_add(__builtin_amdgcn_perm(w.z, w.x, 0x0C060C02u), P2pk);
    CP[3] = pk_add(__builtin_amdgcn_perm(w.z, w.x, 0x0C070C03u), P2pk);
    CP[4] = pk_add(__builtin_amdgcn_perm(w.w, w.y, 0x0C040C00u), P2pk);
    CP[5] = pk_add(__builtin_amdgcn_perm(w.w, w.y, 0x0C050C01u), P2pk);
    CP[6] = pk_add(__builtin_amdgcn_perm(w.w, w.y, 0x0C060C02u), P2pk);
    CP[7] = pk_add(__builtin_amdgcn_perm(w.w, w.y, 0x0C070C03u), P2pk);
}
// private u8 order <-> registers
__device__ __forceinline__ void unpack_p(const uint4 v, uint32_t (&R)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        R[2 * k] = w[k] & 0x00FF00FFu;
        R[2 * k + 1] = __builtin_amdgcn_perm(0u, w[k], SEL_ODD);
    }
}
__device__ __forceinline__ uint4 pack_p(const uint32_t (&R)[8]) {      // values <= 255 per half
    uint4 o;
    o.x = __builtin_amdgcn_perm(R[1], R[0], SEL_PACK);
    o.y = __builtin_amdgcn_perm(R[3], R[2], SEL_PACK);
    o.z = __builtin_amdgcn_perm(R[5], R[4], SEL_PACK);
    o.w = __builtin_amdgcn_perm(R[7], R[6], SEL_PACK);
    return o;
}
// 16 bytes of one of the sweeps' Y volumes (read or written once per pass): FSGM_VOL_NT=1 marks these accesses non-temporal
// (measured: -0.5 to -1.3 %, inside the run-to-run spread: off; with the own columns' C loads marked as well the sweeps
// lose up to 4 % -- the neighbouring workgroup reads the same columns as its halo)
#ifndef FSGM_VOL_NT
#define FSGM_VOL_NT 0
#endif
__device__ __forceinline__ uint4 vol_load(const void* p) { return FSGM_VOL_NT ? load_nt(p) : *(const uint4*)p; }
__device__ __forceinline__ void vol_store(void* p, const uint4 v) { if (FSGM_VOL_NT) store_nt(p, v); else *(uint4*)p = v; }
// The pair kernels' accesses are non-temporal by default: every line of C, Y and the other pair's Y is touched once per pass by
// one wave; measured over three runs each, 8 paths 4.67 -> 4.55 ms per 40 frames, 4 paths 3.64 -> 3.53.  (The sweeps: see vol_load.)
#ifndef FSGM_PAIR_NT
#define FSGM_PAIR_NT 1
#endif
__device__ __forceinline__ uint4 pvol_load(const void* p) { return FSGM_PAIR_NT ? load_nt(p) : *(const uint4*)p; }
__device__ __forceinline__ void pvol_store(void* p, const uint4 v) { if (FSGM_PAIR_NT) store_nt(p, v); else *(uint4*)p = v; }
__device__ __forceinline__ uint4 add4(const uint4 a, const uint4 b) { return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// per-lane constants of the step: the v_perm selectors of the two lane-crossing neighbour registers
struct LaneSel { uint32_t lo, hi; };
template <int LPP>
__device__ __forceinline__ LaneSel lane_sel(const int j) {
    LaneSel s;
    s.lo = j == 0 ? SEL_NB_NOLO : SEL_NB;            // d = 0 has no d-1     (:47)
    s.hi = j == LPP - 1 ? SEL_NB_NOHI : SEL_NB;      // d = D-1 has no d+1   (:48)
    return s;
}

// One DP step (calc_cost_sgm.cpp:33-66) in the mirrored variable, see the header.  S: previous pixel's state
// s = P2 - min(L - m, P2), replaced by the new pixel's; Y: y = P2 - (L_new - C) of the new pixel, in [0, P2].
// A path start (:152-180) = S preset to P2 in every element and mmask = 0 (the stored minimum is 0 there, :154).
template <int LPP, bool MASKED = true>
__device__ __forceinline__ void step_s(uint32_t (&S)[8], const uint32_t (&CP)[8], uint32_t (&Y)[8], const uint32_t P1pk,
                                       const uint32_t P2, const LaneSel sel, const uint32_t mmask) {
    uint32_t T[8], N[8];
    T[7] = pk_subs(S[7], P1pk);                      // the two registers that cross lanes first: the DPP moves below
    T[0] = pk_subs(S[0], P1pk);                      // read them two instructions after they are written
#pragma unroll
    for (int i = 1; i < 7; i++) T[i] = pk_subs(S[i], P1pk);
    // d-1 of register 0 = (previous lane's d = 15, own d = 7); d+1 of register 7 = (own d = 8, next lane's d = 0);
    // lanes without a source lane in their row of 16 read 0
    const uint32_t LT = __builtin_amdgcn_perm(T[7], (uint32_t)__builtin_amdgcn_mov_dpp((int)T[7], DPP_ROW_SHR1, 0xF, 0xF, true), sel.lo);
    const uint32_t RT = __builtin_amdgcn_perm((uint32_t)__builtin_amdgcn_mov_dpp((int)T[0], DPP_ROW_SHL1, 0xF, 0xF, true), T[0], sel.hi);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        Y[i] = pk_max3(S[i], i ? T[i - 1] : LT, i < 7 ? T[i + 1] : RT);
        N[i] = pk_sub(CP[i], Y[i]);
    }
    const uint32_t mm = pk_min(pk_min3(N[0], N[1], N[2]), pk_min3(N[3], N[4], pk_min3(N[5], N[6], N[7])));
    uint32_t mx = group_min_u32<LPP>(min_halves(mm));
    if (MASKED) mx &= mmask;
    const uint32_t p2m = __umul24(mx, 0x10001u) + P2 * 0x10001u;     // (P2 + m) in both halves: one v_mad_u32_u24
#pragma unroll
    for (int i = 0; i < 8; i++) S[i] = pk_subs(p2m, N[i]);
}

// ---------------------------------------------------------------------------------------------
// The step again, for the band sweeps (epi_band.hip), with the instructions that need no packed form as plain 32-bit
// ones -- a 32-bit-encoded v_add_u32 / v_sub_u32 issues in ~3.9 cycles per wave where a v_pk_* takes ~4.8 at four waves
// per SIMD (tools/ubench/pk_rates.hip), and this kernel is bound by vector issue (profiles/r03_sq_counters.md):
//   * y + P1 = max3(s[d] + P1, s[d-1], s[d+1])   -- the clamp of t = max(s - P1, 0) is redundant under a maximum with
//     s[d] >= 0, so in the variable biased by P1 the neighbours are the states themselves and the own term is one plain
//     add; absent neighbours read 0 as before.  The step's Y comes out as y + P1 (YB) and takes its costs biased the
//     same way: CB = C + P2 + P1, so that n = CB - YB = C + P2 - y is unchanged.  Sums of YB carry P1 per path, and
//     paths*CB - sum(YB) = paths*(C + P2) - sum(y) = S: the bias cancels where S is rebuilt.
//   * CB - YB per 32-bit register: no borrow between the halves (n >= 0 in each).
// Needs P1 + P2 <= 127 for the byte forms of the states and sums (epi_band.hip checks).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack_cb(const uint4 w, uint32_t (&CB)[8], const uint32_t biaspk) {   // bias = P2 + P1 in both halves
    CB[0] = __builtin_amdgcn_perm(w.z, w.x, 0x0C040C00u) + biaspk;
    CB[1] = __builtin_amdgcn_perm(w.z, w.x, 0x0C050C01u) + biaspk;
    CB[2] = __builtin_amdgcn_perm(w.z, w.x, 0x0C060C02u) + biaspk;
    CB[3] = __builtin_amdgcn_perm(w.z, w.x, 0x0C070C03u) + biaspk;
    CB[4] = __builtin_amdgcn_perm(w.w, w.y, 0x0C040C00u) + biaspk;
    CB[5] = __builtin_amdgcn_perm(w.w, w.y, 0x0C050C01u) + biaspk;
    CB[6] = __builtin_amdgcn_perm(w.w, w.y, 0x0C060C02u) + biaspk;
    CB[7] = __builtin_amdgcn_perm(w.w, w.y, 0x0C070C03u) + biaspk;
}
template <int LPP, bool MASKED = true>
__device__ __forceinline__ void step_b(uint32_t (&S)[8], const uint32_t (&CB)[8], uint32_t (&YB)[8], const uint32_t P1pk,
                                       const uint32_t P2, const LaneSel sel, const uint32_t mmask) {
    uint32_t N[8];
    // d-1 of register 0 = (previous lane's d = 15, own d = 7); d+1 of register 7 = (own d = 8, next lane's d = 0)
    const uint32_t LT = __builtin_amdgcn_perm(S[7], (uint32_t)__builtin_amdgcn_mov_dpp((int)S[7], DPP_ROW_SHR1, 0xF, 0xF, true), sel.lo);
    const uint32_t RT = __builtin_amdgcn_perm((uint32_t)__builtin_amdgcn_mov_dpp((int)S[0], DPP_ROW_SHL1, 0xF, 0xF, true), S[0], sel.hi);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        YB[i] = pk_max3(S[i] + P1pk, i ? S[i - 1] : LT, i < 7 ? S[i + 1] : RT);
        N[i] = CB[i] - YB[i];
    }
    const uint32_t mm = pk_min(pk_min3(N[0], N[1], N[2]), pk_min3(N[3], N[4], pk_min3(N[5], N[6], N[7])));
    uint32_t mx = group_min_u32<LPP>(min_halves(mm));
    if (MASKED) mx &= mmask;
    const uint32_t p2m = __umul24(mx, 0x10001u) + P2 * 0x10001u;
#pragma unroll
    for (int i = 0; i < 8; i++) S[i] = pk_subs(p2m, N[i]);
}

// hand-off words between workgroups that run at the same time (the chained band sweeps): written and read past
// the L1 and coherently across the XCDs' L2s, 8 bytes at a time; every dword carries its launch's tag in its bytes' top bits
__device__ __forceinline__ uint4 edge_load(const uint4* p) {
    // two 8-byte agent-scope loads: served past the L1 and coherent across the XCDs' L2s
    const unsigned long long lo = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ void edge_store(uint4* p, const uint4 v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((unsigned long long*)p + 1, (unsigned long long)v.z | ((unsigned long long)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// Per-pixel WTA of the final passes: S (packed u16, register layout of the header) of the LPP lanes of a pixel -> one
// record {best, minC, S[best-1], S[best+1]} + S[0] (calc_cost_sgm.cpp:263-271; the parabola runs in
// sweep_finish_kernel).  First minimum over d: inside a lane as packed u16 keys S*16 + (index in the
// lane) (S <= 8*255, so a key fits 16 bits and two of them compare per v_pk_min_u16); across the lanes
// of a pixel as (S << 8 | d).
// sRow (16-byte aligned, 8 dwords per thread of the NT threads that share it; rows private to the wave that writes
// them): TWO PLANES of 16 bytes per lane -- registers 0-3 at dword tid*4, registers 4-7 at dword (NT + tid)*4 -- so that
// every lane's store is one contiguous b128 and a wave's stores cover 1 KiB without a bank conflict (8 dwords per lane
// in one row put lanes k and k+4 on the same banks: measured 0.86 conflict cycles per LDS cycle, profiles/r03_sq_counters.md).
template <int NT>
__device__ __forceinline__ void srow_store(uint32_t* sRow, int tid, const uint32_t (&ST)[8]) {
    *(uint4*)(sRow + tid * 4) = make_uint4(ST[0], ST[1], ST[2], ST[3]);
    *(uint4*)(sRow + (NT + tid) * 4) = make_uint4(ST[4], ST[5], ST[6], ST[7]);
}
// u16 index of element d of the pixel whose first lane is thread tid0: lane d >> 4, register d & 7, half (d >> 3) & 1
template <int NT>
__device__ __forceinline__ uint32_t srow_index(int tid0, uint32_t d) {
    const uint32_t i = d & 7u, t = (uint32_t)tid0 + (d >> 4);
    return ((((i >> 2) * NT + t) * 4 + (i & 3u)) << 1) + ((d >> 3) & 1u);
}

// recb / s0b: byte pointers to the frame's records / S[0] words (wave-uniform), pix: the pixel's index in the frame -- 32-bit
// offsets from uniform bases keep the stores' addresses out of 64-bit vector registers
// STAGE: the record is not stored to HBM here but parked in LDS (stg_rec / stg_s0: this pixel's slot, written by the pixel's
// first lane; an invalid pixel leaves the marker 0xFFFFFFFF) -- the band sweeps' second pass collects eight steps of a row and
// stores them as one 64-byte run (epi_band.hip)
template <int LPP, int NT, bool WTA_MIN3 = false, bool STAGE = false>
__device__ __forceinline__ void wta_row_record_at(const uint32_t (&ST)[8], uint32_t* sRow, int tid, int j,
                                                  bool ok, uint8_t* recb, uint8_t* s0b, uint32_t pix,
                                                  uint2* stg_rec = nullptr, uint16_t* stg_s0 = nullptr) {
    constexpr int D = LPP * 16;
    srow_store<NT>(sRow, tid, ST);
    uint32_t K[8];                                             // keys S*16 + index in the lane: < 2^15, so the fp16 3-input minimum orders them as integers
#pragma unroll
    for (int i = 0; i < 8; i++) K[i] = pk_mad16(ST[i], 0x00100010u, (uint32_t)i | ((uint32_t)(i + 8) << 16));
    const uint32_t kmin = WTA_MIN3 ? pk_min(pk_min3(K[0], K[1], K[2]), pk_min3(K[3], K[4], pk_min3(K[5], K[6], K[7])))
                                   : pk_min(pk_min(pk_min(K[0], K[1]), pk_min(K[2], K[3])), pk_min(pk_min(K[4], K[5]), pk_min(K[6], K[7])));
    const uint32_t k16 = min(kmin & 0xFFFFu, kmin >> 16);
    uint32_t key = ((k16 >> 4) << 8) | ((uint32_t)j * 16u + (k16 & 15u));
    key = group_min_u32<LPP>(key);
    __builtin_amdgcn_wave_barrier();
    if (j == 0 && ok) {
        const uint32_t best = key & 0xFF, minc = key >> 8;
        const uint16_t* srow = (const uint16_t*)sRow;
        const uint32_t c_1 = best > 0 ? srow[srow_index<NT>(tid, best - 1)] : 0u;
        const uint32_t c1 = best + 1 < (uint32_t)D ? srow[srow_index<NT>(tid, best + 1)] : 0u;   // best == D-1: the finish kernel takes the next pixel's S[0]
        if (STAGE) {
            *stg_rec = make_uint2(best | (minc << 16), c_1 | (c1 << 16));
            *stg_s0 = (uint16_t)ST[0];
        } else {
            *(uint2*)(recb + pix * 8u) = make_uint2(best | (minc << 16), c_1 | (c1 << 16));   // (every field below 2^16: sums of <= 16 u8 path costs)
            *(uint16_t*)(s0b + pix * 2u) = (uint16_t)ST[0];        // S[0]: register 0, low half of the pixel's first lane (this one)
        }
    }
}

template <int LPP, int NT>
__device__ __forceinline__ void wta_row_record(const uint32_t (&ST)[8], uint32_t* sRow, int tid, int j,
                                               bool ok, uint2* rec, uint16_t* s0, size_t idx) {
    wta_row_record_at<LPP, NT>(ST, sRow, tid, j, ok, (uint8_t*)(rec + idx), (uint8_t*)(s0 + idx), 0u);
}

}  // namespace

}  // namespace fsgm
