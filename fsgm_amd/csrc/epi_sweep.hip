// epi_sweep.hip -- fused-sweep aggregation for the 8-path calc_cost_sgm case and the pair kernels of the
// 4-path case (gfx950).
//
// Why: the per-direction kernel (epi_kernels.hip, agg_packed_kernel) moves 24 B per voxel through
// HBM (8 reads of C, 8 writes + 8 reads of L_r) and is bandwidth-bound at ~1/5 of the algorithmic
// roofline.  Here the three paths that advance row by row -- from above (0,+1), from above-left
// (+1,+1) and from above-right (-1,+1), calc_cost_sgm.cpp:193-226 -- are computed TOGETHER for the
// same pixel, so C is read once per sweep and only their sum leaves the chip:
//
//   pair kernels (AXIS 0)   C -> Y_h, the two horizontal paths (checkpoint-and-recompute, see the pair section)
//   down sweep   (MODE 0)   C -> Y_dn, the three pass-0 paths from above
//   final sweep  (MODE 2)   on the point-mirrored frame (pass 1): its own three paths, then in registers
//                           S = 8*(C + P2) - (Y_up + Y_dn + Y_h) and the WTA; writes one 10-byte record per PIXEL
//   sweep_finish_kernel     parabola / vz->disparity from the records -> bestD, minC
//
// THE STEP (calc_cost_sgm.cpp:33-66) in the form the kernels compute it.  With L' = L_prev - m_prev the previous
// pixel's path costs less their stored minimum (m = 0 at a path start, :154), the reference's step is
//     x[d] = min(L'[d], min(L'[d-1], L'[d+1]) + P1, P2),   L[d] = C[d] + x[d],   m = min_d L[d]
// when nothing wraps (max C + P2 + max(P1,P2) <= 255, 0 <= P1 <= P2; other parameters use the per-direction
// kernels).  Clamping the state first, Lc = min(L', P2), does not change x (a term above P2 never wins against
// Lc[d] <= P2), and in the mirrored variable  s = P2 - Lc  (0 <= s <= P2)  the step needs four packed
// instructions per register:
//     t[d] = max(s[d] - P1, 0)                      v_pk_sub_u16 clamp
//     y[d] = max(s[d], t[d-1], t[d+1])              v_pk_maximum3_f16   (= P2 - x[d]; absent neighbour: 0)
//     n[d] = (C[d] + P2) - y[d]                     v_pk_sub_u16        (= L[d])
//     s'[d] = max(P2 + m - n[d], 0),  m = min_d n   v_pk_sub_u16 clamp  (= P2 - min(L[d] - m, P2))
// All values are below 1024, i.e. denormal fp16 bit patterns, whose order as fp16 numbers is their order as
// integers: the 3-input fp16 maximum / minimum of gfx950 are exact 3-input u16 maximum / minimum here
// (probed on the device: tools/ubench/pk_rates.hip, and covered by every parity test).  A path start is the same
// instruction stream with s = P2 (x = 0) and the stored minimum masked to 0.
// What leaves the chip are the y of the paths, summed: Y_dn, Y_h (u8), and S is rebuilt as paths*(C + P2) - sum of y.
// Since round 3 the kernels run this step in the P1-biased form of epi_step.h (step_b: y + P1 = max3(s[d] + P1, s[d-1],
// s[d+1]), costs biased by P2 + P1, plain 32-bit adds / subtracts where no packed form is needed): the volumes hold y + P1
// per path (3*(P1+P2) <= 255 for the sweeps, 2*(P1+P2) <= 255 for a pair), and paths*(C + P2 + P1) - sum is the same S.
//
// Lane layout: LPP = D/16 adjacent lanes own a pixel, 16 consecutive d each.  Register i of a lane (i = 0..7) holds
// d = 16j+i in its low and d = 16j+8+i in its high half (u16), so the d-1 / d+1 neighbours of a whole register are
// the whole registers i-1 / i+1 and only registers 0 and 7 need a lane-crossing (one DPP move + one v_perm each).
// The u8 volumes the kernels exchange (Y_dn, Y_h, path states, checkpoints) keep this order -- dword k of a lane's
// 16 bytes = registers 2k (bytes 0, 2) and 2k+1 (bytes 1, 3) -- they are private formats; C and the results are not.
//
// The diagonal paths couple neighbouring columns, so a workgroup that owns a strip of columns needs
// its neighbours' boundary values every row.  Instead of in-kernel neighbour synchronisation the
// sweep is cut into blocks of T rows (one launch per block) and every workgroup recomputes a halo
// of T columns on each side (only the one diagonal direction that flows inwards): a value that is
// wrong because ITS predecessor lay outside the halo moves one column per row and therefore cannot
// reach the strip within T rows.  At block boundaries the three path states of every column go
// through a small global buffer.  No spin-waits, no co-residency assumption.
//
// Workgroup = 4 waves; wave w owns PXW = 64/LPP adjacent columns; strip = 4*PXW columns; T = 2*PXW, so the halo is
// exactly 2 pixel groups per side = one extra unit of work per wave per row (balanced).
// Per row every wave runs 4 DP steps: vertical / both diagonals for its own columns + 1 halo step.
// The diagonal states shift one column per row through LDS (double buffered, one barrier per row).
// MODE 1 (plain up sweep writing Y_up) and wta_sweep_kernel exist for the debug tap that
// rebuilds S in natural order (fsgm_epi_plan_download_sum).
#include "epi_kernels.h"
#include "fsgm_device.h"
#include "epi_wta_tail.h"
#include "epi_step.h"
#include <type_traits>
#include <stdlib.h>

namespace fsgm {

// =============================================================================================
// sweep kernel: rows [y0, y0+rows) of the sweep frame.
//   MODE 0: pass-0 frame, writes Y_dn.   MODE 1: point-mirrored frame, writes Y_up.
//   MODE 2: point-mirrored frame, final: S = 8*(C + P2) - (Y_up + Y_dn + Y_h) in registers,
//           WTA per pixel, writes one record {best, minC, S[best-1], S[best+1]} + S[0] per pixel.
//   MODE 3: the same final on the pass-0 frame (X = Y_up of the rows it covers): the down half of the sweeps that meet
//           in the middle (capi_epi.hip, sweep_mid).
// =============================================================================================
#ifndef FSGM_SWEEP_L16
#define FSGM_SWEEP_L16 1        // diagonal states in LDS as 2 x u16 per dword (A/B knob; 0: packed bytes, half the LDS)
#endif
#ifndef FSGM_SWEEP_MINW
#define FSGM_SWEEP_MINW 1       // minimum waves per SIMD the register allocation must allow (A/B knob)
#endif
#ifndef FSGM_SWEEP_PF
#define FSGM_SWEEP_PF 2         // rows of C in flight per lane in the non-final sweeps (A/B knob; 3: -2 %)
#endif
template <int LPP, int MODE, int NWV, int GPW>
__global__ __launch_bounds__(NWV * 64, FSGM_SWEEP_MINW) void sweep_kernel(SweepArgs a) {
    constexpr bool UP = MODE == 1 || MODE == 2;
    constexpr bool FINAL = MODE >= 2;          // 2: final on the point-mirrored frame, 3: final on the pass-0 frame
    constexpr int PXG = 64 / LPP;            // columns per pixel group (one wave-wide DP step)
    constexpr int PXW = GPW * PXG;           // own columns per wave: GPW groups, each with its three paths, + one halo group
    constexpr int D = LPP * 16;
    constexpr int STRIP = NWV * PXW;         // own columns per workgroup
    constexpr int T = (NWV / 2) * PXG;       // halo width = max rows per launch (one halo step per wave per row)
    // LDS: the diagonal states of one row, as the registers hold them (2 x u16 per dword: no packing on the way), in two
    // planes of 16 bytes per lane (registers 0-3 / 4-7) so that both are conflict-free b128 accesses.  From-above-left
    // states live on columns [a0-T-1, a0+STRIP), from-above-right states on [a0, a0+STRIP+T]: NCD columns each.
    // (L16; the final sweep, whose WTA rows take 8 KB more, and the two-group form keep them packed to bytes in one
    // plane instead -- 12 more instructions per diagonal step, but one more workgroup per CU)
    constexpr int NCD = STRIP + T + 1;
    constexpr bool L16 = !FINAL && GPW == 1 && FSGM_SWEEP_L16 != 0;
    constexpr int PF = FINAL ? 2 : (GPW > 1 ? 2 : FSGM_SWEEP_PF);    // rows of C in flight per lane and group
    __shared__ uint4 sDiag[2][2][L16 ? 2 : 1][NCD * LPP];  // [row parity][direction][plane][column][lane-of-pixel]
    __shared__ __attribute__((aligned(16))) uint32_t sRow[FINAL ? NWV * 64 * 8 : 4];   // final modes: S of the wave's pixels (u16, two planes: epi_step.h)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane / LPP, j = lane % LPP;
    const int W = a.W, H = a.H, NP = W * H;
    // XCD-aware block mapping.  Workgroups are dealt round-robin over the 8 XCDs (each with its own
    // L2), so blocks b, b+8, b+16, ... share an L2.  Give those consecutive (strip, frame) pairs:
    // neighbouring strips re-read each other's C columns and boundary states as halo, and this way
    // those re-reads hit the XCD's L2 instead of going out to the fabric.  Pure speed: any mapping
    // is correct.
    const int nblk = (int)(gridDim.x * gridDim.y);
    const int bid = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;   // bijective (guide T1)
    const int a0 = (lid % (int)gridDim.x) * STRIP;               // first own column of the strip
    const size_t f = lid / (int)gridDim.x;
    const uint8_t* __restrict__ Cf = a.C + f * a.c_frame_stride;
    uint8_t* __restrict__ Xf = a.X + f * a.x_frame_stride;
    const uint8_t* __restrict__ StIn = a.state_in + f * a.state_frame_stride;    // [3][W][D] u8, written by the previous launch
    uint8_t* __restrict__ StOut = a.state_out + f * a.state_frame_stride;        // other buffer: no launch reads what it writes
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2 = (uint32_t)a.P2, P2pk = P2 * 0x10001u;
    const uint32_t Bpk = (P2 + (uint32_t)a.P1) * 0x10001u;       // the costs' bias in the step's variable (epi_step.h, step_b): P2 + P1
    const int y0 = a.y0, rows = min(a.rows, H - y0);
    const bool first_block = y0 == 0;
    const LaneSel sel = lane_sel<LPP>(j);

    // columns in the (possibly mirrored) sweep frame: own group q holds column gx0 + q * PXG
    const int gx0 = a0 + wave * PXW + g;
    // halo unit of this wave: waves 0,1 -> from-above-left on the T columns left of the strip,
    //                         waves 2,3 -> from-above-right on the T columns right of it
    const int hdir = wave >= NWV / 2 ? 1 : 0;
    const int hx = hdir == 0 ? a0 - T + wave * PXG + g : a0 + STRIP + (wave - NWV / 2) * PXG + g;
    const bool halo_ok = hx >= 0 && hx < W;
    const int hxc = min(max(hx, 0), W - 1);
    // A halo column at distance c from the strip feeds the strip's row r + c from its row r: the wave whose nearest halo
    // column is hmin columns away has nothing left to contribute after row rows - 1 - hmin of the block (the far
    // waves' halo is a triangle, not a square)
    const int hmin = hdir == 0 ? T - wave * PXG - PXG + 1 : (wave - NWV / 2) * PXG + 1;
    const int base0 = a0 - T - 1, base1 = a0;                    // forward column of LDS column 0, per direction
    // a diagonal restarts where it enters the image (:156-180): its predecessor column -1 / W holds the start state
    // (never overwritten: stores of columns outside the image are skipped) and the stored minimum is masked to 0
    const uint32_t mask_h = (hdir == 0 ? hx == 0 : hx == W - 1) ? 0u : 0xFFFFu;
    uint32_t mask_dl[GPW], mask_dr[GPW];
    bool plain = mask_h != 0u;
#pragma unroll
    for (int q = 0; q < GPW; q++) {
        mask_dl[q] = gx0 + q * PXG == 0 ? 0u : 0xFFFFu;
        mask_dr[q] = gx0 + q * PXG == W - 1 ? 0u : 0xFFFFu;
        plain = plain && mask_dl[q] != 0u && mask_dr[q] != 0u;
    }
    const bool wave_plain = __all(plain);                        // no lane of this wave ever starts a diagonal below row 0

    auto pix_of = [&](int x, int y) -> int {                     // actual pixel index of sweep-frame (x,y)
        const int p = y * W + x;
        return UP ? NP - 1 - p : p;
    };
    auto vox_off = [&](int x, int y) -> uint32_t {               // byte offset of (x,y)'s 16 bytes of this lane
        return (uint32_t)pix_of(x, y) * D + (uint32_t)j * 16;
    };
    // L16: both planes as ds_read_b128, spelled out -- left to itself the compiler splits the first plane's load into two
    // ds_read2_b32 (dwords 0,3 and 1,2: it wants registers 0 and 7 of the state first), and dword accesses at a 16-byte
    // lane stride put eight lanes on every bank: 0.44 conflict cycles per LDS cycle in round 2's counters
    // (profiles/r02_sq_counters.md; the b128 form reads 0.00 in tools/ubench/counter_calib.hip).  Requests and wait are
    // separate statements so that a row's reads are in flight during its first DP step.
    struct LdsPair { u32x4 lo, hi; };
    auto lds_issue = [&](int par, int dir, int col, LdsPair& v) {
        const uint32_t addr = (uint32_t)(size_t)&sDiag[par][dir][0][col * LPP + j];
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(v.lo), "=&v"(v.hi) : "v"(addr), "n"(NCD * LPP * 16) : "memory");
    };
    auto lds_take = [&](LdsPair& v, uint32_t (&S)[8]) {       // the requests of lds_issue have landed after this
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v.lo), "+v"(v.hi) :: "memory");
        S[0] = v.lo.x; S[1] = v.lo.y; S[2] = v.lo.z; S[3] = v.lo.w; S[4] = v.hi.x; S[5] = v.hi.y; S[6] = v.hi.z; S[7] = v.hi.w;
    };
    auto lds_get = [&](int par, int dir, int col, uint32_t (&S)[8]) {
        if (L16) {
            LdsPair v;
            lds_issue(par, dir, col, v);
            lds_take(v, S);
        } else {
            unpack_p(sDiag[par][dir][0][col * LPP + j], S);
        }
    };
    auto lds_put = [&](int par, int dir, int col, const uint32_t (&S)[8]) {
        if (L16) {
            sDiag[par][dir][0][col * LPP + j] = make_uint4(S[0], S[1], S[2], S[3]);
            sDiag[par][dir][L16 ? 1 : 0][col * LPP + j] = make_uint4(S[4], S[5], S[6], S[7]);
        } else {
            sDiag[par][dir][0][col * LPP + j] = pack_p(S);
        }
    };

    // ---- block prologue: path states of the row above (from the previous launch; start states above row 0) ----
    uint32_t VS[GPW][8];
#pragma unroll
    for (int q = 0; q < GPW; q++) {
        if (first_block) {
#pragma unroll
            for (int i = 0; i < 8; i++) VS[q][i] = P2pk;
        } else {
            unpack_p(*(const uint4*)(StIn + (size_t)min(gx0 + q * PXG, W - 1) * D + j * 16), VS[q]);
        }
    }
    for (int i = tid; i < 2 * NCD * LPP; i += NWV * 64) {
        const int dir = i / (NCD * LPP), r = i - dir * (NCD * LPP);
        const int c = r / LPP, jj = r - c * LPP;
        const int x = (dir ? base1 : base0) + c;
        const bool inside = x >= 0 && x < W;
        uint4 pk = make_uint4(P2 * 0x01010101u, P2 * 0x01010101u, P2 * 0x01010101u, P2 * 0x01010101u);   // the start state, packed
        if (inside && !first_block) pk = *(const uint4*)(StIn + ((size_t)(1 + dir) * W + x) * D + jj * 16);
        if (L16) {
            uint32_t S[8];
            unpack_p(pk, S);
            const uint4 lo = make_uint4(S[0], S[1], S[2], S[3]), hi = make_uint4(S[4], S[5], S[6], S[7]);
            sDiag[0][dir][0][r] = lo; sDiag[0][dir][L16 ? 1 : 0][r] = hi;
            sDiag[1][dir][0][r] = lo; sDiag[1][dir][L16 ? 1 : 0][r] = hi;     // columns outside the image keep the start state in both buffers
        } else {
            sDiag[0][dir][0][r] = pk;
            sDiag[1][dir][0][r] = pk;
        }
    }
    __syncthreads();

    const uint8_t* __restrict__ Lhf = FINAL ? a.Lh + f * a.lh_frame_stride : nullptr;
    // rows in flight per lane: C of the own groups and of the halo group; MODE 2: the other partial sums of the own
    // pixels (Y_dn, Y_h) as well -- all requested PF rows ahead, so no row waits for HBM
    uint4 ringOwn[GPW][PF], ringHalo[PF], ringX[FINAL ? GPW : 1][PF], ringH[FINAL ? GPW : 1][PF];
#pragma unroll
    for (int i = 0; i < PF; i++) {
        const int y = min(y0 + i, H - 1);
#pragma unroll
        for (int q = 0; q < GPW; q++) {
            const uint32_t off = vox_off(min(gx0 + q * PXG, W - 1), y);
            ringOwn[q][i] = *(const uint4*)(Cf + off);
            if (FINAL) { ringX[q][i] = vol_load(Xf + off); ringH[q][i] = vol_load(Lhf + off); }
        }
        ringHalo[i] = *(const uint4*)(Cf + vox_off(hxc, y));
    }

    // one row of the block: 3 * GPW + 1 DP steps per wave, one barrier.  MASKED: a row in which a path of this wave
    // starts (row 0; waves at an image border): the stored minima go through their masks.
    auto do_row = [&](const int k, const uint4 (&cOwn)[GPW], const uint4 cHalo, const uint4 (&curX)[GPW], const uint4 (&curH)[GPW], auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int y = y0 + k, par = k & 1;
        const uint32_t tmask = y == 0 ? 0u : 0xFFFFu;            // row 0: every path starts (:152-180)
#pragma unroll
        for (int q = 0; q < GPW; q++) {
            const int gx = gx0 + q * PXG;
            const bool own_ok = gx < W;
            uint32_t CP[8], Y[8], YS[8];
            LdsPair pdl, pdr;
            if (L16) { lds_issue(par, 0, gx - 1 - base0, pdl); lds_issue(par, 1, gx + 1 - base1, pdr); }    // in flight during the first step
            unpack_cb(cOwn[q], CP, Bpk);

            // from above (0,+1)                                            calc_cost_sgm.cpp:193-202
            step_b<LPP, MASKED>(VS[q], CP, YS, P1pk, P2, sel, tmask);

            // from above-left (+1,+1): predecessor column gx-1                        :205-213
            {
                uint32_t S[8];
                if (L16) lds_take(pdl, S); else lds_get(par, 0, gx - 1 - base0, S);
                step_b<LPP, MASKED>(S, CP, Y, P1pk, P2, sel, mask_dl[q] & tmask);
                if (own_ok) lds_put(par ^ 1, 0, gx - base0, S);
#pragma unroll
                for (int i = 0; i < 8; i++) YS[i] += Y[i];
            }
            // from above-right (-1,+1): predecessor column gx+1                       :215-225
            {
                uint32_t S[8];
                if (L16) lds_take(pdr, S); else lds_get(par, 1, gx + 1 - base1, S);
                step_b<LPP, MASKED>(S, CP, Y, P1pk, P2, sel, mask_dr[q] & tmask);
                if (own_ok) lds_put(par ^ 1, 1, gx - base1, S);
#pragma unroll
                for (int i = 0; i < 8; i++) YS[i] += Y[i];
            }
            if (!FINAL) {
                // sum of this sweep's three y, one byte per voxel (3*P2 <= 255)      :227-232
                if (own_ok) vol_store(Xf + vox_off(gx, y), pack_p(YS));
            } else {
                // S = 8*(C + P2) - (this sweep's Y (registers) + the other sweep's Y + Y_h), all at this pixel
                uint32_t E2[8], ST[8];
                unpack_p(curX[q], E2);
#pragma unroll
                for (int i = 0; i < 8; i++) YS[i] += E2[i];
                if (a.lh_natural) unpack_c(curH[q], E2, 0u);         // Y_h of the pairx_* kernels: natural d order
                else unpack_p(curH[q], E2);
#pragma unroll
                for (int i = 0; i < 8; i++) ST[i] = pk_sub(pk_mad16(CP[i], 0x00080008u, 0u), pk_add(YS[i], E2[i]));
                wta_row_record<LPP, NWV * 64>(ST, sRow, tid, j, own_ok, a.rec, a.s0, f * (size_t)NP + pix_of(min(gx, W - 1), y));
            }
        }
        // halo unit: keeps the inward-flowing diagonal correct for the next rows
        if (k + hmin < rows) {                                   // wave-uniform
            uint32_t HP[8], S[8], Y[8];
            unpack_cb(cHalo, HP, Bpk);
            const int px = hdir == 0 ? hx - 1 : hx + 1;
            lds_get(par, hdir, px - (hdir ? base1 : base0), S);
            step_b<LPP, MASKED>(S, HP, Y, P1pk, P2, sel, mask_h & tmask);
            if (halo_ok) lds_put(par ^ 1, hdir, hx - (hdir ? base1 : base0), S);
        }
        __syncthreads();                                         // diagonal states of row y visible to row y+1
    };
    auto row = [&](const int k, const uint4 (&cOwn)[GPW], const uint4 cHalo, const uint4 (&cX)[GPW], const uint4 (&cH)[GPW]) {
        if (wave_plain && y0 + k > 0) do_row(k, cOwn, cHalo, cX, cH, std::false_type{});      // wave-uniform
        else do_row(k, cOwn, cHalo, cX, cH, std::true_type{});
    };

    // steady state without branches so the prefetched rows stay in flight; then the tail
    int k0 = 0;
    for (; k0 + PF <= rows; k0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            uint4 cOwn[GPW], cX[GPW], cH[GPW];
            const uint4 cHalo = ringHalo[i];
            const int yn = min(y0 + k0 + i + PF, H - 1);
#pragma unroll
            for (int q = 0; q < GPW; q++) {
                const uint32_t off = vox_off(min(gx0 + q * PXG, W - 1), yn);
                cOwn[q] = ringOwn[q][i];
                ringOwn[q][i] = *(const uint4*)(Cf + off);
                if (FINAL) {
                    cX[q] = ringX[q][i]; cH[q] = ringH[q][i];
                    ringX[q][i] = vol_load(Xf + off); ringH[q][i] = vol_load(Lhf + off);
                }
            }
            ringHalo[i] = *(const uint4*)(Cf + vox_off(hxc, yn));
            row(k0 + i, cOwn, cHalo, cX, cH);
        }
    }
#pragma unroll
    for (int i = 0; i < PF - 1; i++)
        if (k0 + i < rows) {                                     // block-uniform
            uint4 cOwn[GPW], cX[GPW], cH[GPW];
#pragma unroll
            for (int q = 0; q < GPW; q++) { cOwn[q] = ringOwn[q][i]; if (FINAL) { cX[q] = ringX[q][i]; cH[q] = ringH[q][i]; } }
            row(k0 + i, cOwn, ringHalo[i], cX, cH);
        }

    // ---- block epilogue: states of the last row for the next launch ----
    if (y0 + rows < H) {
        const int par = rows & 1;                                // buffer the last row wrote into
#pragma unroll
        for (int q = 0; q < GPW; q++) {
            const int gx = gx0 + q * PXG;
            if (gx < W) {
                uint32_t S[8];
                *(uint4*)(StOut + (size_t)gx * D + j * 16) = pack_p(VS[q]);
                lds_get(par, 0, gx - base0, S);
                *(uint4*)(StOut + ((size_t)W + gx) * D + j * 16) = pack_p(S);
                lds_get(par, 1, gx - base1, S);
                *(uint4*)(StOut + ((size_t)2 * W + gx) * D + j * 16) = pack_p(S);
            }
        }
    }
}

// =============================================================================================
// finish kernel for MODE 2: parabola + vz->disparity from the per-pixel records
// (calc_cost_sgm.cpp:278-308, :414-426).  best == D-1 reads the next pixel's S[0] (:296).
// =============================================================================================
__global__ __launch_bounds__(256) void sweep_finish_kernel(WtaArgs a, const uint2* __restrict__ rec,
                                                           const uint16_t* __restrict__ s0) {
    const int NP = a.W * a.H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const size_t f = blockIdx.y;
    const uint2 r = rec[f * (size_t)NP + p];
    const uint32_t best = r.x & 0xFFFFu, minc = r.x >> 16, c_1 = r.y & 0xFFFFu;
    uint32_t c1 = r.y >> 16;
    if (best + 1 == (uint32_t)a.D) c1 = p + 1 < NP ? (uint32_t)s0[f * (size_t)NP + p + 1] : 0u;
    wta_finish(a, f, p, best, minc, c_1, c1);
}

// =============================================================================================
// WTA over S = nC*(C + P2) - (Y_dn + Y_up + Y_h)  (calc_cost_sgm.cpp:227-232, :259-308, :414-426)
// for the non-final mode (Y_up materialised by a MODE 1 sweep): the debug tap that rebuilds S in natural order.
// One thread per pixel-lane as in the sweeps; the Y volumes are in the private byte order (header).
// =============================================================================================
// the four volumes wta_sweep_kernel adds up are read once: past the caches (0.336 -> 0.321 ms for 8 frames, three runs each)
#ifndef FSGM_WTAS_NT
#define FSGM_WTAS_NT 1
#endif
__device__ __forceinline__ uint4 wvol_load(const void* p) { return FSGM_WTAS_NT ? load_nt(p) : *(const uint4*)p; }

template <int LPP>
__global__ __launch_bounds__(256) void wta_sweep_kernel(WtaArgs a, SweepSumArgs q) {
    constexpr int D = LPP * 16;
    constexpr int PPB = 256 / LPP;
    __shared__ __attribute__((aligned(16))) uint32_t sS[256 * 8];     // two planes of 16 bytes per lane (epi_step.h, srow_store)
    const int tid = threadIdx.x;
    const int NP = a.W * a.H;
    const int gp = blockIdx.x * PPB + tid / LPP, j = tid % LPP;
    const bool valid = gp < NP;
    const int p = valid ? gp : NP - 1;
    const size_t f = blockIdx.y;
    const size_t bo = (size_t)p * D + (size_t)j * 16;                    // byte offset in a u8 volume
    const uint32_t Bpk = (uint32_t)q.bias * 0x10001u, nC = (uint32_t)q.nC;     // bias = P2 + P1: the Y volumes hold y + P1 per path (step_b)
    uint32_t CP[8], YT[8], E2[8], ST[8];
    unpack_cb(wvol_load(q.C + f * q.v_frame_stride + bo), CP, Bpk);
    unpack_p(wvol_load(q.Xdn + f * q.v_frame_stride + bo), YT);
    if (q.Xup) {
        unpack_p(wvol_load(q.Xup + f * q.v_frame_stride + bo), E2);
#pragma unroll
        for (int k = 0; k < 8; k++) YT[k] += E2[k];
    }
    if (q.Lx) {                                                          // the along-x paths as two path volumes (line kernels)
        uint32_t L0[8];
        unpack_c(wvol_load(q.Lx + f * q.lx_frame_stride + bo), L0, 0u);
        unpack_c(wvol_load(q.Lx + f * q.lx_frame_stride + (q.lx_frame_stride >> 1) + bo), E2, 0u);
#pragma unroll
        for (int k = 0; k < 8; k++) ST[k] = pk_add(pk_sub(pk_mad16(CP[k], nC * 0x10001u, 0u), YT[k]), pk_add(L0[k], E2[k]));
    } else {
        if (q.lh_natural) unpack_c(wvol_load(q.Lh + f * q.lh_frame_stride + bo), E2, 0u);
        else unpack_p(wvol_load(q.Lh + f * q.lh_frame_stride + bo), E2);
#pragma unroll
        for (int k = 0; k < 8; k++) ST[k] = pk_sub(pk_mad16(CP[k], nC * 0x10001u, 0u), pk_add(YT[k], E2[k]));
    }

    uint32_t key = 0xFFFFFFFFu;
    srow_store<256>(sS, tid, ST);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t v0 = ST[k] & 0xFFFF, v1 = ST[k] >> 16;
        const uint32_t d0 = (uint32_t)j * 16 + k, d1 = d0 + 8;
        key = min(key, min((v0 << 8) | d0, (v1 << 8) | d1));
        if (q.Sdbg && valid) {
            uint32_t* o = q.Sdbg + f * (size_t)NP * D + (size_t)p * D;
            o[d0] = v0; o[d1] = v1;
        }
    }
    key = group_min_u32<LPP>(key);
    __syncthreads();
    if (j == 0 && valid) {
        const uint32_t best = key & 0xFF, minc = key >> 8;
        const uint16_t* srow = (const uint16_t*)sS;
        uint32_t c_1 = 0, c1 = 0;
        if (a.subpixel && best > 1) {
            c_1 = srow[srow_index<256>(tid, best - 1)];
            if (best + 1 < (uint32_t)D) c1 = srow[srow_index<256>(tid, best + 1)];
            else if (p + 1 < NP) {                                           // next pixel's d=0 (:296): byte 0 of its lane 0 in every volume
                const size_t nb = f * q.v_frame_stride + (size_t)(p + 1) * D;
                const size_t nh = f * q.lh_frame_stride + (size_t)(p + 1) * D;
                c1 = nC * ((uint32_t)q.C[nb] + (uint32_t)q.bias) - ((uint32_t)q.Xdn[nb] + (q.Xup ? (uint32_t)q.Xup[nb] : 0u));
                if (q.Lx) { const size_t nx = f * q.lx_frame_stride + (size_t)(p + 1) * D; c1 += (uint32_t)q.Lx[nx] + (uint32_t)q.Lx[nx + (q.lx_frame_stride >> 1)]; }
                else c1 -= (uint32_t)q.Lh[nh];
            }
        }
        wta_finish(a, f, p, best, minc, c_1, c1);
    }
}

// =============================================================================================
// An opposite pair of paths as ONE sum  Y = y_fwd + y_bwd  (<= 2*P2, one byte).
// The two paths of an axis (calc_cost_sgm.cpp:183-202 and their pass-1 mirrors) run in opposite
// directions along a line, so their values for a pixel exist at different times; writing both path
// volumes and reading them back costs 6 B per voxel (C twice, 2 writes, 2 reads).
// Checkpoint-and-recompute brings that to ~4.25 B:
//   pass A (pair_ckpt_kernel)  end -> start of the line, keeps nothing but the backward
//          state at every HP_TC-th position (1/HP_TC B per voxel);
//   pass B (pair_sum_kernel)   start -> end in tiles of HP_TC positions: the tile's backward y
//          are recomputed from the checkpoint on its far edge into registers, then the forward path
//          crosses the tile, adds them and stores Y.  The tile's C stays in registers between the two.
// AXIS 0: lines = image rows (the horizontal pair; 64/LPP rows per wave, consecutive positions 16*LPP
//         bytes apart); AXIS 1: lines = image columns (the vertical pair; 64/LPP adjacent columns per
//         wave -- one contiguous run per step -- consecutive positions a row apart).
// FINAL:  instead of storing Y, pass B adds the other axis' Y and does the WTA on the spot --
//         the 4-path pipeline (the reference's shipped configuration): horizontal pair -> Y_h, vertical
//         pair final: S = 4*(C + P2) - (Y_v + Y_h), 7.5 B per voxel for 4 voxel-paths, S never in HBM.
// =============================================================================================
#ifndef FSGM_PAIR_PF
#define FSGM_PAIR_PF 4          // steps of C the checkpoint passes request ahead (A/B knob)
#endif
#ifndef FSGM_HP_TC
#define FSGM_HP_TC 8            // tile width = checkpoint spacing in positions (A/B knob)
#endif
constexpr int HP_TC = FSGM_HP_TC;

template <int LPP, int AXIS>
__global__ __launch_bounds__(256) void pair_ckpt_kernel(PairArgs a) {
    constexpr int PXW = 64 / LPP, D = LPP * 16, PF = FSGM_PAIR_PF;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPP, j = lane % LPP;
    const int nl = AXIS ? a.W : a.H, len = AXIS ? a.H : a.W;
    const int lg = (int)blockIdx.x * 4 + wave;
    if (lg * PXW >= nl) return;                                 // wave-uniform
    // the pair kernels are few waves with a long serial chain each (len steps): beside the sweeps' many waves they
    // must not queue for issue slots, or the whole stage waits for them
    __builtin_amdgcn_s_setprio(3);
    const int l = min(lg * PXW + g, nl - 1);                    // lines past the last redo the last (same bytes, same addresses)
    const int NT = (len + HP_TC - 1) / HP_TC;
    if (NT < 2) return;                                         // a single tile starts at the border: no checkpoint
    const size_t f = blockIdx.y;
    const size_t lstride = AXIS ? (size_t)D : (size_t)a.W * D, tstride = AXIS ? (size_t)a.W * D : (size_t)D;
    const uint8_t* __restrict__ Cl = a.C + f * a.c_frame_stride + (size_t)l * lstride + (size_t)j * 16;
    uint8_t* __restrict__ Kl = a.ckpt + f * a.ckpt_frame_stride + ((size_t)l * (NT - 1)) * D + (size_t)j * 16;
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2 = (uint32_t)a.P2, P2pk = P2 * 0x10001u;
    const uint32_t Bpk = (P2 + (uint32_t)a.P1) * 0x10001u;       // the costs' bias in the step's variable (epi_step.h, step_b): P2 + P1
    const LaneSel sel = lane_sel<LPP>(j);
    uint32_t S[8];
#pragma unroll
    for (int i = 0; i < 8; i++) S[i] = P2pk;                    // the path starts at the line's last position
    auto load_c = [&](int t) -> uint4 { return pvol_load(Cl + (size_t)max(t, 0) * tstride); };
    uint4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) ring[i] = load_c(len - 1 - i);
    const int last = HP_TC;                                     // the nearest checkpoint to the line start
    for (int t0 = len - 1; t0 >= last; t0 -= PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const int t = t0 - i;
            const uint4 cw = ring[i];
            ring[i] = load_c(t - PF);
            uint32_t CP[8], Y[8];
            unpack_cb(cw, CP, Bpk);
            step_b<LPP>(S, CP, Y, P1pk, P2, sel, t == len - 1 ? 0u : 0xFFFFu);
            // positions below `last` in the final group are computed but not needed; t >= 0 always holds there
            if (t >= last && (t % HP_TC) == 0) *(uint4*)(Kl + (size_t)(t / HP_TC - 1) * D) = pack_p(S);
        }
    }
}

template <int LPP, int AXIS, bool FINAL>
__global__ __launch_bounds__(256, 2) void pair_sum_kernel(PairArgs a) {        // two waves per SIMD: at most 256 registers
    constexpr int PXW = 64 / LPP, D = LPP * 16, TC = HP_TC;
    __shared__ __attribute__((aligned(16))) uint32_t sRow[FINAL ? 4 * 64 * 8 : 4];            // FINAL: S of the wave's pixels (u16)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane / LPP, j = lane % LPP;
    const int W = a.W, NP = a.W * a.H;
    const int nl = AXIS ? a.W : a.H, len = AXIS ? a.H : a.W;
    const int lg = (int)blockIdx.x * 4 + wave;
    if (lg * PXW >= nl) return;
    __builtin_amdgcn_s_setprio(3);
    const bool own_ok = lg * PXW + g < nl;
    const int l = min(lg * PXW + g, nl - 1);
    const int NT = (len + TC - 1) / TC;
    const size_t f = blockIdx.y;
    const size_t lstride = AXIS ? (size_t)D : (size_t)W * D, tstride = AXIS ? (size_t)W * D : (size_t)D;
    const size_t lbase = (size_t)l * lstride + (size_t)j * 16;
    const uint8_t* __restrict__ Cl = a.C + f * a.c_frame_stride + lbase;
    uint8_t* __restrict__ Xl = FINAL ? nullptr : a.X + f * a.x_frame_stride + lbase;     // !FINAL: this pair's sum, out
    const uint8_t* __restrict__ Ol = FINAL ? a.Xother + f * a.xo_frame_stride + lbase : nullptr;   // FINAL: the other pair's, in
    const uint8_t* __restrict__ Kl = a.ckpt + f * a.ckpt_frame_stride + ((size_t)l * max(NT - 1, 1)) * D + (size_t)j * 16;
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2 = (uint32_t)a.P2, P2pk = P2 * 0x10001u;
    const uint32_t Bpk = (P2 + (uint32_t)a.P1) * 0x10001u;       // the costs' bias in the step's variable (epi_step.h, step_b): P2 + P1
    const LaneSel sel = lane_sel<LPP>(j);
    uint32_t FS[8];                                               // forward state, carried across tiles
#pragma unroll
    for (int i = 0; i < 8; i++) FS[i] = P2pk;                    // position 0 starts the forward path
    auto load_c = [&](int t) -> uint4 { return pvol_load(Cl + (size_t)min(t, len - 1) * tstride); };
    auto load_k = [&](int t) -> uint4 { return *(const uint4*)(Kl + (size_t)min(t, max(NT - 2, 0)) * D); };
    uint4 cT[TC], cN[TC], kT = load_k(0), kN;
#pragma unroll
    for (int c = 0; c < TC; c++) cT[c] = load_c(c);
    // one tile; EDGE: the tile holds position 0 or reaches the line's end, where the paths start (:152-180) -- the
    // other tiles run without a single select
    auto tile = [&](const int t, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        const int tb = t * TC;
#pragma unroll
        for (int c = 0; c < TC; c++) cN[c] = load_c(tb + TC + c);      // next tile, in flight while this one computes
        kN = load_k(t + 1);
        // backward path through the tile: positions past the line end come first and are wiped by the path
        // start at the last position; a tile inside the line starts from its checkpoint
        uint32_t RS[8];
        unpack_p(kT, RS);
        uint4 exR[TC];
#pragma unroll
        for (int c = TC - 1; c >= 0; c--) {
            const int x = tb + c;
            uint32_t CP[8], Y[8];
            unpack_cb(cT[c], CP, Bpk);
            uint32_t mmask = 0xFFFFu;
            if (EDGE && x >= len - 1) {                           // wave-uniform: the backward path starts here
#pragma unroll
                for (int i = 0; i < 8; i++) RS[i] = P2pk;
                mmask = 0u;
            }
            step_b<LPP>(RS, CP, Y, P1pk, P2, sel, mmask);
            exR[c] = pack_p(Y);
        }
        uint4 xo[TC];
        if (FINAL) {
#pragma unroll
            for (int c = 0; c < TC; c++) xo[c] = pvol_load(Ol + (size_t)min(tb + c, len - 1) * tstride);
        }
        // forward path, adding the two y
#pragma unroll
        for (int c = 0; c < TC; c++) {
            const int x = tb + c;
            uint32_t CP[8], Y[8];
            unpack_cb(cT[c], CP, Bpk);
            step_b<LPP>(FS, CP, Y, P1pk, P2, sel, (EDGE && x == 0) ? 0u : 0xFFFFu);
            if (!FINAL) {
                // both y are <= P2 per byte and 2*P2 <= 255: the packed bytes add as plain words
                if (!EDGE || x < len) pvol_store(Xl + (size_t)x * tstride, add4(pack_p(Y), exR[c]));
            } else {
                // S = nC*(C + P2) - (this pair + the other pair) (calc_cost_sgm.cpp:227-232), WTA on the spot
                uint32_t ST[8], E2[8], E3[8];
                unpack_p(exR[c], E2);
                if (a.xo_natural) unpack_c(xo[c], E3, 0u); else unpack_p(xo[c], E3);    // (block-uniform)
                const uint32_t nC = (uint32_t)a.nC * 0x10001u;
#pragma unroll
                for (int q = 0; q < 8; q++) ST[q] = pk_sub(pk_mad16(CP[q], nC, 0u), pk_add(pk_add(Y[q], E2[q]), E3[q]));
                const int ap = AXIS ? x * W + l : l * W + x;     // pixel index (wave-uniform validity: x < len)
                wta_row_record<LPP, 256>(ST, sRow, tid, j, own_ok && (!EDGE || x < len), a.rec, a.s0, f * (size_t)NP + (size_t)min(ap, NP - 1));
            }
        }
#pragma unroll
        for (int c = 0; c < TC; c++) cT[c] = cN[c];
        kT = kN;
    };
    if (NT == 1) {
        tile(0, std::true_type{});
    } else {
        tile(0, std::true_type{});                                // holds position 0 (and, for len <= TC + 1, the end too)
        for (int t = 1; t < NT - 1; t++) {
            if ((t + 1) * TC <= len - 1) tile(t, std::false_type{});
            else tile(t, std::true_type{});
        }
        tile(NT - 1, std::true_type{});
    }
}

// =============================================================================================
// The along-x pair with 8 costs a lane (half the registers of the kernels above: 4 packed registers, register i =
// (d = 8j + i, d = 8j + 4 + i), D / 8 lanes a pixel, 64 / (D / 8) rows a wave).  The along-x lines are few and long:
// where nothing else runs beside them (the 4-path pipeline) their serial length is what everything waits for, and
// a step of a lane with half the costs is ≈0.6 of the instructions while twice the waves share the SIMDs.
// Same passes, same checkpoint spacing; the checkpoint bytes and the Y of the tile stay in a private order of this
// pair of kernels, the stored sum Y is in natural d order (PairArgs.xo_natural / SweepSumArgs.lh_natural tell its readers).
// =============================================================================================
__device__ __forceinline__ void unpack_c4(const uint2 w, uint32_t (&CP)[4], const uint32_t biaspk) {     // bias = P2 + P1 (step_b's variable)
    CP[0] = __builtin_amdgcn_perm(w.y, w.x, 0x0C040C00u) + biaspk;
    CP[1] = __builtin_amdgcn_perm(w.y, w.x, 0x0C050C01u) + biaspk;
    CP[2] = __builtin_amdgcn_perm(w.y, w.x, 0x0C060C02u) + biaspk;
    CP[3] = __builtin_amdgcn_perm(w.y, w.x, 0x0C070C03u) + biaspk;
}
// private byte order of 4 registers: (R0.lo, R1.lo, R0.hi, R1.hi), (R2.lo, R3.lo, R2.hi, R3.hi)
__device__ __forceinline__ uint2 pack_q(const uint32_t (&R)[4]) {
    return make_uint2(__builtin_amdgcn_perm(R[1], R[0], SEL_PACK), __builtin_amdgcn_perm(R[3], R[2], SEL_PACK));
}
__device__ __forceinline__ void unpack_q(const uint2 v, uint32_t (&R)[4]) {
    R[0] = v.x & 0x00FF00FFu; R[1] = __builtin_amdgcn_perm(0u, v.x, SEL_ODD);
    R[2] = v.y & 0x00FF00FFu; R[3] = __builtin_amdgcn_perm(0u, v.y, SEL_ODD);
}
// private order -> natural d order: (d0 d1 d4 d5), (d2 d3 d6 d7) -> (d0 d1 d2 d3), (d4 d5 d6 d7)
__device__ __forceinline__ uint2 natural_q(const uint2 v) {
    return make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x05040100u), __builtin_amdgcn_perm(v.y, v.x, 0x07060302u));
}

// step_b (epi_step.h: the step in the P1-biased variable) for 4 registers a lane, G lanes a pixel (G <= 16: one DPP row)
template <int G, bool MASKED = true>
__device__ __forceinline__ void step_q(uint32_t (&S)[4], const uint32_t (&CB)[4], uint32_t (&YB)[4], const uint32_t P1pk,
                                       const uint32_t P2, const LaneSel sel, const uint32_t mmask) {
    uint32_t N[4];
    const uint32_t LT = __builtin_amdgcn_perm(S[3], (uint32_t)__builtin_amdgcn_mov_dpp((int)S[3], DPP_ROW_SHR1, 0xF, 0xF, true), sel.lo);
    const uint32_t RT = __builtin_amdgcn_perm((uint32_t)__builtin_amdgcn_mov_dpp((int)S[0], DPP_ROW_SHL1, 0xF, 0xF, true), S[0], sel.hi);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        YB[i] = pk_max3(S[i] + P1pk, i ? S[i - 1] : LT, i < 3 ? S[i + 1] : RT);
        N[i] = CB[i] - YB[i];
    }
    const uint32_t mm = pk_min(N[0], pk_min3(N[1], N[2], N[3]));
    uint32_t mx = group_min_u32<G>(min_halves(mm));
    if (MASKED) mx &= mmask;
    const uint32_t p2m = __umul24(mx, 0x10001u) + P2 * 0x10001u;
#pragma unroll
    for (int i = 0; i < 4; i++) S[i] = pk_subs(p2m, N[i]);
}

template <int D>
__global__ __launch_bounds__(256) void pairx_ckpt_kernel(PairArgs a) {
    constexpr int G = D / 8, PXW = 64 / G, PF = FSGM_PAIR_PF;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, j = lane % G;
    const int nl = a.H, len = a.W;
    const int lg = (int)blockIdx.x * 4 + wave;
    if (lg * PXW >= nl) return;                                 // wave-uniform
    const int l = min(lg * PXW + g, nl - 1);                    // lines past the last redo the last (same bytes, same addresses)
    const int NT = (len + HP_TC - 1) / HP_TC;
    if (NT < 2) return;
    const size_t f = blockIdx.y;
    const uint8_t* __restrict__ Cl = a.C + f * a.c_frame_stride + (size_t)l * a.W * D + (size_t)j * 8;
    uint8_t* __restrict__ Kl = a.ckpt + f * a.ckpt_frame_stride + ((size_t)l * (NT - 1)) * D + (size_t)j * 8;
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2 = (uint32_t)a.P2, P2pk = P2 * 0x10001u;
    const uint32_t Bpk = (P2 + (uint32_t)a.P1) * 0x10001u;       // the costs' bias in the step's variable (epi_step.h, step_b): P2 + P1
    const LaneSel sel = lane_sel<G>(j);
    uint32_t S[4] = {P2pk, P2pk, P2pk, P2pk};                   // the path starts at the line's last position
    auto load_c = [&](int t) -> uint2 { return *(const uint2*)(Cl + (size_t)max(t, 0) * D); };
    uint2 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) ring[i] = load_c(len - 1 - i);
    const int last = HP_TC;
    for (int t0 = len - 1; t0 >= last; t0 -= PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const int t = t0 - i;
            const uint2 cw = ring[i];
            ring[i] = load_c(t - PF);
            uint32_t CP[4], Y[4];
            unpack_c4(cw, CP, Bpk);
            step_q<G>(S, CP, Y, P1pk, P2, sel, t == len - 1 ? 0u : 0xFFFFu);
            if (t >= last && (t % HP_TC) == 0) *(uint2*)(Kl + (size_t)(t / HP_TC - 1) * D) = pack_q(S);
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void pairx_sum_kernel(PairArgs a) {
    constexpr int G = D / 8, PXW = 64 / G, TC = HP_TC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, j = lane % G;
    const int nl = a.H, len = a.W;
    const int lg = (int)blockIdx.x * 4 + wave;
    if (lg * PXW >= nl) return;
    const int l = min(lg * PXW + g, nl - 1);
    const int NT = (len + TC - 1) / TC;
    const size_t f = blockIdx.y;
    const size_t lbase = (size_t)l * a.W * D + (size_t)j * 8;
    const uint8_t* __restrict__ Cl = a.C + f * a.c_frame_stride + lbase;
    uint8_t* __restrict__ Xl = a.X + f * a.x_frame_stride + lbase;
    const uint8_t* __restrict__ Kl = a.ckpt + f * a.ckpt_frame_stride + ((size_t)l * max(NT - 1, 1)) * D + (size_t)j * 8;
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2 = (uint32_t)a.P2, P2pk = P2 * 0x10001u;
    const uint32_t Bpk = (P2 + (uint32_t)a.P1) * 0x10001u;       // the costs' bias in the step's variable (epi_step.h, step_b): P2 + P1
    const LaneSel sel = lane_sel<G>(j);
    uint32_t FS[4] = {P2pk, P2pk, P2pk, P2pk};                  // position 0 starts the forward path
    auto load_c = [&](int t) -> uint2 { return *(const uint2*)(Cl + (size_t)min(t, len - 1) * D); };
    auto load_k = [&](int t) -> uint2 { return *(const uint2*)(Kl + (size_t)min(t, max(NT - 2, 0)) * D); };
    uint2 cT[TC], cN[TC], kT = load_k(0), kN;
#pragma unroll
    for (int c = 0; c < TC; c++) cT[c] = load_c(c);
    auto tile = [&](const int t, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        const int tb = t * TC;
#pragma unroll
        for (int c = 0; c < TC; c++) cN[c] = load_c(tb + TC + c);
        kN = load_k(t + 1);
        uint32_t RS[4];
        unpack_q(kT, RS);
        uint2 exR[TC];
#pragma unroll
        for (int c = TC - 1; c >= 0; c--) {
            const int x = tb + c;
            uint32_t CP[4], Y[4];
            unpack_c4(cT[c], CP, Bpk);
            uint32_t mmask = 0xFFFFu;
            if (EDGE && x >= len - 1) {                           // wave-uniform: the backward path starts here
                RS[0] = RS[1] = RS[2] = RS[3] = P2pk;
                mmask = 0u;
            }
            step_q<G>(RS, CP, Y, P1pk, P2, sel, mmask);
            exR[c] = pack_q(Y);
        }
#pragma unroll
        for (int c = 0; c < TC; c++) {
            const int x = tb + c;
            uint32_t CP[4], Y[4];
            unpack_c4(cT[c], CP, Bpk);
            step_q<G>(FS, CP, Y, P1pk, P2, sel, (EDGE && x == 0) ? 0u : 0xFFFFu);
            const uint2 yf = pack_q(Y);
            // both y are <= P2 per byte and 2*P2 <= 255: the packed bytes add as plain words
            if (!EDGE || x < len) *(uint2*)(Xl + (size_t)x * D) = natural_q(make_uint2(yf.x + exR[c].x, yf.y + exR[c].y));
        }
#pragma unroll
        for (int c = 0; c < TC; c++) cT[c] = cN[c];
        kT = kN;
    };
    if (NT == 1) {
        tile(0, std::true_type{});
    } else {
        tile(0, std::true_type{});
        for (int t = 1; t < NT - 1; t++) {
            if ((t + 1) * TC <= len - 1) tile(t, std::false_type{});
            else tile(t, std::true_type{});
        }
        tile(NT - 1, std::true_type{});
    }
}

// =============================================================================================
// Self-test of the step's 3-input maximum / minimum (pk_max3 / pk_min3): they are exact u16 operations only because
// (a) hipcc fuses the nested 2-input builtins into v_pk_maximum3_f16 / v_pk_minimum3_f16 -- or emits anything else
// that keeps denormal inputs -- and (b) the kernel's float mode does not flush fp16 denormals.  A toolchain or flag
// change that breaks either would give silently wrong path costs; one wave checks 64 x 32 triples of values below
// 1024 in both halves (every value the step produces) against integer max / min when the first plan of a device is made.
// =============================================================================================
__global__ void pk3_selftest_kernel(uint32_t* bad) {
    uint32_t x = 0x9E3779B9u * (threadIdx.x + 1u), nbad = 0;
    for (int it = 0; it < 32; it++) {
        uint32_t v[3];
        for (int k = 0; k < 3; k++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t lo = (x >> 7) & 1023u, hi = (x >> 19) & 1023u;
            // the corner values in the first rounds: 0, 1, the largest denormal
            v[k] = it == 0 ? (k == 0 ? 0u : k == 1 ? 0x00010001u : 0x03FF03FFu) : (lo | (hi << 16));
        }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));      // not foldable at compile time
        const uint32_t mx = pk_max3(v[0], v[1], v[2]), mn = pk_min3(v[0], v[1], v[2]);
        const uint32_t rmx = max(max(v[0] & 0xFFFFu, v[1] & 0xFFFFu), v[2] & 0xFFFFu) | (max(max(v[0] >> 16, v[1] >> 16), v[2] >> 16) << 16);
        const uint32_t rmn = min(min(v[0] & 0xFFFFu, v[1] & 0xFFFFu), v[2] & 0xFFFFu) | (min(min(v[0] >> 16, v[1] >> 16), v[2] >> 16) << 16);
        nbad += (mx != rmx) + (mn != rmn);
    }
    if (nbad) atomicAdd(bad, nbad);
}

// 0 = the fused step's packed 3-input operations are exact on this device / build; > 0 = mismatches; < 0 = HIP error
int fused_step_selftest(hipStream_t st) {
    uint32_t* d = nullptr;
    uint32_t h = 0;
    if (hipMalloc((void**)&d, sizeof(uint32_t)) != hipSuccess) return -1;
    hipError_t e = hipMemsetAsync(d, 0, sizeof(uint32_t), st);
    if (e == hipSuccess) { hipLaunchKernelGGL(pk3_selftest_kernel, dim3(1), dim3(64), 0, st, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    return e == hipSuccess ? (int)h : -1;
}

size_t pair_ckpt_bytes(int W, int H, int D, int axis) {
    const int len = axis ? H : W, nl = axis ? W : H;
    const int NT = (len + HP_TC - 1) / HP_TC;
    return (size_t)nl * (size_t)(NT > 1 ? NT - 1 : 1) * D;
}

// LDS the along-y checkpoint pass asks for without using it: a cap of two resident workgroups per CU.  In the 4-path
// pipeline that pass (many waves, short lines) runs beside the along-x pair (few waves, the 2 W-step chain everything waits
// for) and crowds its waves out of the issue slots: capped, 40 frames take 3.47 instead of 3.62 ms.  FSGM_PAIR_VLDS (KB): A/B knob.
static size_t pair_vckpt_lds() {
    static const size_t v = [] { const char* e = getenv("FSGM_PAIR_VLDS"); const int x = (e && *e) ? atoi(e) : 64; return (size_t)(x < 0 ? 0 : (x > 160 ? 160 : x)) * 1024; }();
    return v;
}

template <int LPP>
static void launch_pair_t(hipStream_t st, const PairArgs& a, int frames, int axis, bool final_pass, int phase) {
    constexpr int PXW = 64 / LPP;
    const int nl = axis ? a.W : a.H;
    dim3 grid((nl + 4 * PXW - 1) / (4 * PXW), frames);
    if (phase != 2) {
        if (axis == 0) hipLaunchKernelGGL((pair_ckpt_kernel<LPP, 0>), grid, dim3(256), 0, st, a);
        else           hipLaunchKernelGGL((pair_ckpt_kernel<LPP, 1>), grid, dim3(256), pair_vckpt_lds(), st, a);
    }
    if (phase != 1) {
        if (axis == 0 && !final_pass)      hipLaunchKernelGGL((pair_sum_kernel<LPP, 0, false>), grid, dim3(256), 0, st, a);
        else if (axis == 0)                hipLaunchKernelGGL((pair_sum_kernel<LPP, 0, true>), grid, dim3(256), 0, st, a);
        else if (!final_pass)              hipLaunchKernelGGL((pair_sum_kernel<LPP, 1, false>), grid, dim3(256), 0, st, a);
        else                               hipLaunchKernelGGL((pair_sum_kernel<LPP, 1, true>), grid, dim3(256), 0, st, a);
    }
}

// the along-x pair as pairx_* (8 costs a lane): D = 16 .. 128; Y comes out in natural d order
bool pair_x_fine_ok(int D) { return D == 16 || D == 32 || D == 64 || D == 128; }
void launch_pair_x_fine(hipStream_t st, const PairArgs& a, int frames) {
#define FSGM_PX(DD) do { constexpr int PXW = 64 / (DD / 8); dim3 grid((a.H + 4 * PXW - 1) / (4 * PXW), frames); \
        hipLaunchKernelGGL((pairx_ckpt_kernel<DD>), grid, dim3(256), 0, st, a); \
        hipLaunchKernelGGL((pairx_sum_kernel<DD>), grid, dim3(256), 0, st, a); } while (0)
    switch (a.D) {
        case 16: FSGM_PX(16); break;
        case 32: FSGM_PX(32); break;
        case 64: FSGM_PX(64); break;
        case 128: FSGM_PX(128); break;
        default: break;
    }
#undef FSGM_PX
}

// One axis (0 horizontal, 1 vertical).  phase 0: checkpoint pass + sum pass; 1: checkpoint pass only;
// 2: sum pass only (so that the caller can put an event between them).  final_pass: the sum pass adds
// a.Xother and writes WTA records instead of Y.
void launch_pair(hipStream_t st, const PairArgs& a0, int frames, int axis, bool final_pass, int phase) {
    const PairArgs& a = a0;
    switch (agg_packed_lpp(a.D)) {
        case 1: launch_pair_t<1>(st, a, frames, axis, final_pass, phase); break;
        case 2: launch_pair_t<2>(st, a, frames, axis, final_pass, phase); break;
        case 4: launch_pair_t<4>(st, a, frames, axis, final_pass, phase); break;
        case 8: launch_pair_t<8>(st, a, frames, axis, final_pass, phase); break;
        case 16: launch_pair_t<16>(st, a, frames, axis, final_pass, phase); break;
        default: break;
    }
}

// =============================================================================================
// launchers
// =============================================================================================
#ifndef FSGM_SWEEP_WAVES
#define FSGM_SWEEP_WAVES 4      // waves per sweep workgroup (4: 32-column strips, 16 rows per launch at D = 128)
#endif
int sweep_rows_per_launch(int D) { const int lpp = agg_packed_lpp(D); return lpp ? (FSGM_SWEEP_WAVES / 2) * (64 / lpp) : 0; }
size_t sweep_state_bytes(int W, int D) { return (size_t)3 * W * D; }

// (sweep_kernel's GPW parameter: pixel groups per wave.  2 -- six own DP steps and one halo step per row instead of three and one,
// 64-column strips -- halves the halo's weight but needs twice the registers and leaves half the workgroups: measured 7 % slower
// (4.04 vs 3.78 ms per 32 frames); only GPW = 1 is instantiated.)
// rows [ybeg, yend) of the sweep frame, T a launch; *parity: which of the two state buffers the next launch writes (carried from
// one range of a sweep to the next)
template <int LPP, int GPW, int NWV>
static void launch_sweep_g(hipStream_t st, SweepArgs a, int frames, int mode, int ybeg, int yend, int* parity) {
    constexpr int STRIP = NWV * GPW * (64 / LPP), T = (NWV / 2) * (64 / LPP);
    dim3 grid((a.W + STRIP - 1) / STRIP, frames);
    uint8_t* const buf0 = a.state_out;                     // caller passes the base of 2 x frames x state buffers
    uint8_t* const buf1 = a.state_out + (size_t)frames * a.state_frame_stride;
    int b = *parity;
    for (int y0 = ybeg; y0 < yend; y0 += T, b ^= 1) {
        a.y0 = y0;
        a.rows = std::min(T, yend - y0);
        a.state_in = b ? buf0 : buf1;
        a.state_out = b ? buf1 : buf0;
        if (mode == 0)      hipLaunchKernelGGL((sweep_kernel<LPP, 0, NWV, GPW>), grid, dim3(NWV * 64), 0, st, a);
        else if (mode == 1) hipLaunchKernelGGL((sweep_kernel<LPP, 1, NWV, GPW>), grid, dim3(NWV * 64), 0, st, a);
        else if (mode == 2) hipLaunchKernelGGL((sweep_kernel<LPP, 2, NWV, GPW>), grid, dim3(NWV * 64), 0, st, a);
        else                hipLaunchKernelGGL((sweep_kernel<LPP, 3, NWV, GPW>), grid, dim3(NWV * 64), 0, st, a);
    }
    *parity = b;
}

// tall = 1: workgroups of 8 waves (64-column strips, 32 rows per launch at D = 128) for the non-final modes -- half the
// launches of a sweep.  A sweep of a small batch is a chain of launches that no other work hides (parallel sweeps, mode 3):
// fewer, longer launches shorten it; large batches keep the 4-wave form (more workgroups per CU, round 1's measurement).
template <int LPP>
static void launch_sweep_t(hipStream_t st, SweepArgs a, int frames, int mode, int tall, int ybeg, int yend, int* parity) {
    if (tall) launch_sweep_g<LPP, 1, 8>(st, a, frames, mode, ybeg, yend, parity);
    else launch_sweep_g<LPP, 1, FSGM_SWEEP_WAVES>(st, a, frames, mode, ybeg, yend, parity);
}

void launch_sweep_rows(hipStream_t st, const SweepArgs& a, int frames, int mode, int tall, int ybeg, int yend, int* parity) {
    switch (agg_packed_lpp(a.D)) {
        case 1: launch_sweep_t<1>(st, a, frames, mode, tall, ybeg, yend, parity); break;
        case 2: launch_sweep_t<2>(st, a, frames, mode, tall, ybeg, yend, parity); break;
        case 4: launch_sweep_t<4>(st, a, frames, mode, tall, ybeg, yend, parity); break;
        case 8: launch_sweep_t<8>(st, a, frames, mode, tall, ybeg, yend, parity); break;
        case 16: launch_sweep_t<16>(st, a, frames, mode, tall, ybeg, yend, parity); break;
        default: break;
    }
}

void launch_sweep(hipStream_t st, const SweepArgs& a, int frames, int mode, int tall) {
    int parity = 0;
    launch_sweep_rows(st, a, frames, mode, tall, 0, a.H, &parity);
}

void launch_sweep_finish(hipStream_t st, const WtaArgs& a, const uint2* rec, const uint16_t* s0, int frames) {
    hipLaunchKernelGGL(sweep_finish_kernel, dim3((a.W * a.H + 255) / 256, frames), dim3(256), 0, st, a, rec, s0);
}

void launch_wta_sweep(hipStream_t st, const WtaArgs& a, const SweepSumArgs& q, int frames) {
    const int NP = a.W * a.H;
    const int lpp = agg_packed_lpp(a.D);
    dim3 grid((NP + 256 / lpp - 1) / (256 / lpp), frames);
    switch (lpp) {
        case 1: hipLaunchKernelGGL(wta_sweep_kernel<1>, grid, dim3(256), 0, st, a, q); break;
        case 2: hipLaunchKernelGGL(wta_sweep_kernel<2>, grid, dim3(256), 0, st, a, q); break;
        case 4: hipLaunchKernelGGL(wta_sweep_kernel<4>, grid, dim3(256), 0, st, a, q); break;
        case 8: hipLaunchKernelGGL(wta_sweep_kernel<8>, grid, dim3(256), 0, st, a, q); break;
        case 16: hipLaunchKernelGGL(wta_sweep_kernel<16>, grid, dim3(256), 0, st, a, q); break;
    }
}

}  // namespace fsgm
