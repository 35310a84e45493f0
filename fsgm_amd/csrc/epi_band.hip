// epi_band.hip -- band sweeps: ALL FOUR paths of a raster pass of calc_cost_sgm.cpp:86-257 in one sweep (gfx950).
//
// Why.  The block sweeps of epi_sweep.hip compute the three paths that advance row by row together and leave the two
// along-x paths to the pair kernels: C is read four times, the pair's sum Y_h is written and read, a quarter of the sweeps'
// DP steps are halo recomputation and the pair recomputes one of its two paths -- 9.7 B per voxel through HBM and
// ~880 wave-instructions per 8 pixels for 8 voxel-paths each.  A raster pass's four paths -- from the left (-1,0), from
// above (0,-1), from above-left (-1,-1), from above-right (+1,-1), calc_cost_sgm.cpp:183-226 -- all take their
// predecessor from an earlier pixel in raster order, so they CAN be computed together if the sweep order respects all four
// dependencies: in the skewed coordinate u = x + 2*y the predecessors of (u, y) are (u-1, y), (u-2, y-1), (u-3, y-1) and
// (u-1, y-1) -- all at smaller u.  A workgroup therefore walks u and computes, per step, one pixel of every row of a BAND of
// rows, all four paths each: C is read once per pass, the only volume between the two passes is the sum of the first
// pass's four y (9 bits per voxel: a byte volume plus a bit plane when 4*P2 > 255), and there is no halo, no recomputed path,
// no pair kernel:
//
//   down pass (MODE 0)   C -> Y_dn (+ bit plane), the four pass-0 paths
//   final pass (MODE 2)  on the point-mirrored frame (pass 1, :114-123): its own four paths, then in registers
//                        S = 8*(C + P2) - (Y_up + Y_dn) and the WTA; one 10-byte record per pixel (sweep_finish_kernel
//                        does the parabola / vz conversion as for the block sweeps)
//   5.3 B per voxel through HBM (PMC counters, profiles/r03_pmc_traffic.json) instead of 9.7, ~650 wave-instructions per
//   8 pixels instead of ~880.
//
// PATHS = 4 (the shipped configuration, :104): the two paths of a pass that remain (from the left, from above) need only
// u = x + y; Y_dn <= 2*P2 fits a byte; 4.5 B per voxel (measured) instead of 7.5.
//
// Layout.  LPP = D/16 lanes own a pixel (16 costs each, register layout of epi_step.h); a wave owns 64/LPP consecutive
// ROWS (lane group g <-> row), a workgroup of NWV waves a band of R = NWV*64/LPP rows; one workgroup walks one frame,
// band after band (375 rows = 6 bands of 64).  The from-the-left state never leaves its registers.  The three states that
// cross rows go through LDS: every row writes its new states of step u into buffer u&1 at its own row slot and reads the
// row above's states of step u-1 from the other buffer -- that is pixel x+1 of the row above: the from-above-right
// predecessor; the from-above (pixel x) and from-above-left (pixel x-1) predecessors are the same reads of one and two steps
// ago, kept in registers (packed bytes, 12 VGPRs).  One workgroup barrier per step.  Between bands the last row's states
// go through a per-frame global buffer indexed by image column, written in place behind the reads (a band reads column
// u+2 of it at step u and writes column u-2(R-1)); wave 0 plays "row -1" for row 0.  Frames are independent workgroups: no
// inter-workgroup synchronisation of any kind, no tickets, no polling.
//
// Path starts (:152-180): a predecessor outside the image means the path starts here -- the consumer presets the state to
// P2 in every element and masks the stored minimum to 0 (epi_step.h); waves whose eight pixels are all strictly inside
// the image run a variant without those selects.
//
// What it needs: many frames.  One band of one frame is in flight per workgroup, two workgroups of 8 waves per CU: 512
// frames fill the chip.  That is what 288 GB of HBM are for; smaller batches take the block sweeps / line kernels.
#include "epi_kernels.h"
#include "fsgm_device.h"
#include "epi_step.h"
#include <type_traits>
#include <stdlib.h>

namespace fsgm {

#ifndef FSGM_BAND_PF
#define FSGM_BAND_PF 2          // steps of C in flight per lane, first pass (A/B knob)
#endif
#ifndef FSGM_BAND_PF2
#define FSGM_BAND_PF2 1         // steps of C, Y_dn and its bit plane in flight per lane, final pass (A/B knob; 2 spills registers inside the loop: 57.4 -> 45.8 ms per 512 frames with 1)
#endif
#ifndef FSGM_BAND_PF4
#define FSGM_BAND_PF4 3         // the same for both passes of the 4-path form (A/B knob; 1 / 2 / 3 / 4: 31.5 / 30.0 / 28.9 / 29.1 ms per 512 frames)
#endif
#ifndef FSGM_BAND_PFE
#define FSGM_BAND_PFE 4         // chained form: steps of hand-off words in flight, first pass (A/B knob)
#endif
#ifndef FSGM_BAND_PFE2
#define FSGM_BAND_PFE2 2        // the same, second pass (registers)
#endif
#ifndef FSGM_BAND_R16
#define FSGM_BAND_R16 1         // first pass: the from-above-right state as 2 x u16 in LDS (A/B knob)
#endif
#ifndef FSGM_BAND_Y16
#define FSGM_BAND_Y16 0         // 1: the first pass's sums beyond a byte (4*(P1+P2) > 255) cross to the second pass as the registers hold
                                // them (2 x u16: two 16-byte planes per lane) instead of low bytes + a 9th-bit plane -- 0.75 B per voxel
                                // more each way for 11 + 16 fewer instructions per step.  Measured in round 4 (512 frames, same box,
                                // alternating): SQ_INSTS_VALU -4.4 % / -3.4 % per pass as intended, durations 18.70 / 24.43 ms against
                                // 18.77 / 23.62 -- the chip clocks DOWN with the extra bytes (GRBM cycles / duration: 2.09 against
                                // 2.15 GHz in the first pass): the stage runs into the power limit, where instructions traded for
                                // bytes buy nothing.  Off; kept as a knob (profiles/r04_band_y16.txt)
#endif
#ifndef FSGM_BAND_RECLDS
#define FSGM_BAND_RECLDS 1      // second pass: the 10-byte WTA records of a wave's rows collect in LDS for eight steps and leave as 64-byte
                                // runs (8-byte stores scattered over a wave's 8 rows were written to HBM 6.7 times over: profiles/r03_pmc_traffic.json)
#endif
#ifndef FSGM_BAND_SLACK
#define FSGM_BAND_SLACK 32      // chained form: columns of lead a band gives the band above before it starts (A/B knob)
#endif
#ifndef FSGM_BAND_WAVES
#define FSGM_BAND_WAVES 8        // waves per band workgroup (8: 64-row bands at D = 128, two workgroups per CU)
#endif
#ifndef FSGM_BAND_MINW
#define FSGM_BAND_MINW 4         // waves per SIMD the register allocation must allow (A/B knob: 3 with 6-wave workgroups)
#endif

// 9th bits of the eight packed-u16 registers of a lane -> one dword (bit k of byte b = bit 8 of register 2k + (b&1), half b>>1)
__device__ __forceinline__ uint32_t pack_hi_bits(const uint32_t (&R)[8]) {
    constexpr uint32_t SEL_HI = 0x07030501u;                 // v_perm(b, a): bytes a.1, b.1, a.3, b.3
    const uint32_t h0 = __builtin_amdgcn_perm(R[1], R[0], SEL_HI), h1 = __builtin_amdgcn_perm(R[3], R[2], SEL_HI);
    const uint32_t h2 = __builtin_amdgcn_perm(R[5], R[4], SEL_HI), h3 = __builtin_amdgcn_perm(R[7], R[6], SEL_HI);
    return h0 | (h1 << 1) | (h2 << 2) | (h3 << 3);
}
// private u8 order + bit plane -> registers (values up to 511)
__device__ __forceinline__ void unpack_p9(const uint4 v, const uint32_t bits, uint32_t (&R)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t hk = (bits >> k) & 0x01010101u;
        R[2 * k] = __builtin_amdgcn_perm(hk, w[k], 0x06020400u);          // (w.0, h.0, w.2, h.2)
        R[2 * k + 1] = __builtin_amdgcn_perm(hk, w[k], 0x07030501u);      // (w.1, h.1, w.3, h.3)
    }
}

template <int LPP, int MODE, int NWV, int PATHS, bool BITS, bool TAP, bool CHAIN>
__global__ __launch_bounds__(NWV * 64, FSGM_BAND_MINW) void band_kernel(BandArgs a) {
    constexpr bool UP = MODE != 0;
    constexpr bool P8 = PATHS == 8;
    constexpr int PXG = 64 / LPP;            // rows per wave
    constexpr int R = NWV * PXG;             // rows per band
    constexpr int D = LPP * 16;
    constexpr int SKEW = P8 ? 2 : 1;         // u = x + SKEW * (row in band)
    constexpr int NST = P8 ? 3 : 1;          // states that cross rows: 0 from above, 1 from above-left, 2 from above-right
    constexpr int PF = P8 ? (MODE == 2 ? FSGM_BAND_PF2 : FSGM_BAND_PF) : FSGM_BAND_PF4;   // 4 paths: half the registers, twice the bytes per instruction
    // first pass, 8 paths: the from-above-right state -- written and read every step, never held -- crosses rows as the registers
    // hold it (2 x u16 per dword, two planes of 16 bytes per lane): no pack on the way in, no unpack on the way out, for 17 KB
    // more LDS (67 KB: still two workgroups per CU; the second pass needs that room for its WTA rows)
    constexpr bool R16 = P8 && MODE == 0 && !CHAIN && FSGM_BAND_R16 != 0;   // (the chained form has no registers to spare for it)
    constexpr int NSL = R16 ? 2 : NST;                       // states that cross rows as packed bytes
    __shared__ uint4 sR16[R16 ? 2 : 1][2][R16 ? (R + 1) * LPP : 1];   // [step parity][plane: registers 0-3 / 4-7][row slot][lane of pixel]
    __shared__ uint4 sSt[2][NSL][(R + 1) * LPP];              // [step parity][state][row slot (row + 1; slot 0 = the row above the band)][lane of pixel]
    __shared__ __attribute__((aligned(16))) uint32_t sRow[MODE == 2 ? NWV * 64 * 8 : 4];   // final pass: S of the wave's pixels (u16, two planes: epi_step.h)
    constexpr bool RECLDS = MODE == 2 && FSGM_BAND_RECLDS != 0 && LPP == 8;   // (8 lanes a pixel: a wave's 64 lanes = 8 rows x 8 steps of records)
    __shared__ uint2 sRec[RECLDS ? NWV * 64 : 1];              // [wave][row of the wave][step & 7]
    __shared__ uint16_t sRs0[RECLDS ? NWV * 64 : 1];
    __shared__ uint32_t sTicket;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane / LPP, j = lane % LPP;
    const int r = wave * PXG + g;                              // row within the band
    const int W = a.W, H = a.H, NP = W * H;
    // CHAIN: this workgroup walks ONE band of one frame; the bands of a frame run at the same time as workgroups of their own
    // and hand their last row's states over while they run (below).  (band, frame) come from a ticket counter, band-major
    // within a group of frames, so the band above always holds an earlier ticket: it is running or done, whatever the dispatch
    // order or placement.
    int cband = 0;
    size_t f = blockIdx.x;
    if (CHAIN) {
        if (tid == 0) sTicket = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const uint32_t t = __builtin_amdgcn_readfirstlane(sTicket);
        const uint32_t per = (uint32_t)a.group * (uint32_t)a.nbands, grp = t / per, rem = t - grp * per;
        const uint32_t fin = min((uint32_t)a.group, (uint32_t)a.frames - grp * (uint32_t)a.group);    // frames in this group
        cband = (int)(rem / fin);
        f = (size_t)grp * a.group + rem % fin;
        if (cband >= a.nbands || f >= (size_t)a.frames) return;                                        // (never: the grid is frames x bands)
    }
    const uint8_t* __restrict__ Cf = a.C + f * a.c_frame_stride;
    uint8_t* __restrict__ Yf = a.Y + f * a.y_frame_stride;
    constexpr bool Y16 = BITS && FSGM_BAND_Y16 != 0;         // BITS: the second volume is registers 4-7 as they are (Yf: registers 0-3), not a bit plane
    uint8_t* __restrict__ Bf = BITS ? (uint8_t*)(a.Yb + f * a.yb_frame_stride) : nullptr;     // [NP][LPP] dwords (Y16: [NP][LPP] uint4)
    uint8_t* __restrict__ Ef = (uint8_t*)(a.edge + f * a.edge_frame_stride);                  // [W][NST][LPP] uint4 (CHAIN: one such map per band boundary)
    const uint32_t tag = CHAIN ? a.tag : 0u;
    uint8_t* __restrict__ recb = MODE == 2 ? (uint8_t*)(a.rec + f * (size_t)NP) : nullptr;
    uint8_t* __restrict__ s0b = MODE == 2 ? (uint8_t*)(a.s0 + f * (size_t)NP) : nullptr;
    // every global access below is a wave-uniform base + a 32-bit byte offset per lane (no 64-bit address registers)
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2 = (uint32_t)a.P2, P2pk = P2 * 0x10001u;
    const uint32_t Bpk = (P2 + (uint32_t)a.P1) * 0x10001u;     // the costs' bias in the step's variable (epi_step.h, step_b): P2 + P1
    const uint4 startP = make_uint4(P2 * 0x01010101u, P2 * 0x01010101u, P2 * 0x01010101u, P2 * 0x01010101u);
    const LaneSel sel = lane_sel<LPP>(j);
    const int elane = min(lane, NST * LPP - 1);                // wave 0: lane = state * LPP + lane-of-pixel of the hand-off words

    auto pix_of = [&](int x, int y) -> int { const int p = y * W + x; return UP ? NP - 1 - p : p; };

    for (int yb = CHAIN ? cband * R : 0; yb < (CHAIN ? cband * R + 1 : H); yb += R) {   // ---- one band (CHAIN: this workgroup's only one) ----
        const int Rp = min(R, H - yb);
        const int y = yb + r, yc = min(y, H - 1);
        const bool row_ok = r < Rp;
        const int nsteps = W + SKEW * (Rp - 1);
        const bool have_above = yb > 0, last_band = yb + R >= H;
        const bool loader = wave == 0 && lane < NST * LPP && have_above;
        // steps in which this wave has a pixel inside the image / in which all of its pixels are strictly inside
        const int r_lo = wave * PXG, r_hi = r_lo + PXG - 1;
        // (8 paths: one step early -- the from-above / from-above-left predecessors of a row's first pixels are read from
        // LDS one and two steps before they are used, and the slot they sit in is rewritten every other step)
        const int act_lo = SKEW * r_lo - (P8 ? 1 : 0), act_hi = W - 1 + SKEW * r_hi;
        const bool wave_rows = r_lo < Rp;
        const int pl_lo = SKEW * r_hi + 1, pl_hi = W - 2 + SKEW * r_lo;
        const bool wave_plain_rows = r_hi < Rp && yb + r_lo >= 1;

        auto vox_off = [&](int u) -> uint32_t {               // byte offset of this lane's 16 bytes at step u (clamped into the image)
            const int x = min(max(u - SKEW * r, 0), W - 1);
            return (uint32_t)pix_of(x, yc) * D + (uint32_t)j * 16;
        };
        auto bit_off = [&](int u) -> uint32_t {                // byte offset of this lane's dword of the bit plane
            const int x = min(max(u - SKEW * r, 0), W - 1);
            return ((uint32_t)pix_of(x, yc) * LPP + (uint32_t)j) * 4u;
        };
        // hand-off maps of this band: the one it reads (written by the band above) and the one it writes
        const uint32_t emap = (uint32_t)W * (NST * LPP) * 16u;
        const uint8_t* __restrict__ Ein = CHAIN ? Ef + (uint32_t)max(cband - 1, 0) * emap : Ef;
        uint8_t* __restrict__ Eout = CHAIN ? Ef + (uint32_t)cband * emap : Ef;
        auto edge_at = [&](int x) -> const uint8_t* { return Ein + ((uint32_t)min(max(x, 0), W - 1) * (NST * LPP) + (uint32_t)elane) * 16u; };
        // CHAIN: a hand-off word is this launch's when every dword carries the launch's tag in its bytes' top bits (states are
        // below 128); the band above may not have got there yet: poll, bounded, and raise a.err instead of hanging
        bool gave_up = false;                                  // (wave 0) a hand-off wait timed out: results are invalid, finish without waiting
        // wave 0 as "row -1": the band above's three states of one column into row slot 0 of buffer `par`
        auto put_above = [&](const int par, const uint4 v) {
            if (R16 && elane / LPP == 2) {
                uint32_t S[8];
                unpack_p(v, S);
                sR16[par][0][elane % LPP] = make_uint4(S[0], S[1], S[2], S[3]);
                sR16[par][1][elane % LPP] = make_uint4(S[4], S[5], S[6], S[7]);
            } else {
                sSt[par][elane / LPP][elane % LPP] = v;
            }
        };
        auto fresh = [&](const uint4 v) -> bool { return (((v.x ^ tag) | (v.y ^ tag) | (v.z ^ tag) | (v.w ^ tag)) & 0x80808080u) == 0u; };
        auto eload = [&](const uint8_t* q) -> uint4 { return CHAIN ? edge_load((const uint4*)q) : load_nt(q); };
        auto settle = [&](uint4 v, const uint8_t* q, const bool mine) -> uint4 {       // mine: this lane takes part in the hand-off
            if (CHAIN) {
                // a wait of 2^18 polls (a good fraction of a second; the longest legitimate one is a few milliseconds: the band
                // above is 2R columns ahead by construction) gives up for good -- this workgroup and, through a.err, every other
                for (uint32_t spins = 0; !gave_up && !__all(!mine || fresh(v)); spins++) {
                    __builtin_amdgcn_s_sleep(8);
                    if (mine) v = edge_load((const uint4*)q);
                    if (spins > (1u << 18) || ((spins & 1023u) == 1023u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        if (lane == 0) atomicOr(a.err, 1u);
                        gave_up = true;
                    }
                }
                v.x &= 0x7F7F7F7Fu; v.y &= 0x7F7F7F7Fu; v.z &= 0x7F7F7F7Fu; v.w &= 0x7F7F7F7Fu;
            }
            return v;
        };

        uint32_t FS[8];                                        // from the left: stays in these lanes
#pragma unroll
        for (int i = 0; i < 8; i++) FS[i] = P2pk;
        // from above / from above-left of the row above, read one / two steps before they are used
        uint4 hU = startP, hD1 = startP, hD2 = startP;
        if (P8 && have_above && wave == 0) {                   // row 0: column 0 of the band above ("read at step -1"); wave-uniform
            const bool mine = r == 0;
            const uint8_t* q0 = Ein + (uint32_t)(0 * LPP + j) * 16u;
            const uint8_t* q1 = Ein + (uint32_t)(1 * LPP + j) * 16u;
            const uint4 v0 = settle(mine ? eload(q0) : startP, q0, mine), v1 = settle(mine ? eload(q1) : startP, q1, mine);
            if (mine) { hU = v0; hD1 = v1; }
        }
        if (CHAIN && wave == 0 && have_above) {
            // Start only once the band above is FSGM_BAND_SLACK columns further than the first step needs: both bands then
            // advance at the same rate with that much slack between them, and a step's hand-off words are there when they are
            // asked for (without it the two run in lock step and every hiccup of the band above is a poll down here).
            const uint8_t* q = edge_at(min(SKEW + FSGM_BAND_SLACK, W - 1));
            (void)settle(loader ? eload(q) : startP, q, loader);
        }
        if (wave == 0 && have_above) {                         // what row 0 reads at step 0
            const uint8_t* q = edge_at(SKEW - 1);
            const uint4 v = settle(loader ? eload(q) : startP, q, loader);
            if (loader) put_above(1, v);
        }
        uint4 ringC[PF], ringY[MODE == 2 ? PF : 1];
        // wave 0: the band above's states for the coming steps.  Sequential form: one step ahead (the lines were written a band
        // earlier: L2 / Infinity Cache).  Chained form: the words were written moments ago by another CU with write-through stores
        // and come from beyond the L2 -- a round trip of about a step's time -- so several steps are kept in flight.
        constexpr int PFE = CHAIN ? (MODE == 2 ? FSGM_BAND_PFE2 : FSGM_BAND_PFE) : 1;
        uint4 ringE[PFE];
#pragma unroll
        for (int i = 0; i < PFE; i++) ringE[i] = startP;
        uint32_t ringB[MODE == 2 && BITS && !Y16 ? PF : 1];
        uint4 ringB4[MODE == 2 && Y16 ? PF : 1];
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const uint32_t off = vox_off(i);
            ringC[i] = *(const uint4*)(Cf + off);
            if (MODE == 2) {
                ringY[i] = vol_load(Yf + off);
                if (Y16) ringB4[i] = vol_load(Bf + off);
                else if (BITS) ringB[i] = *(const uint32_t*)(Bf + bit_off(i));
            }
        }
        if (loader) {
#pragma unroll
            for (int i = 0; i < PFE; i++) ringE[i] = eload(edge_at(SKEW + i));
        }
        if constexpr (RECLDS) sRec[wave * 64 + lane] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        __syncthreads();

        // one step of this wave's rows.  EDGE: a pixel of the wave is at / outside an image border or in row 0, or a row
        // of the wave lies below the image (selects allowed); the plain variant has none.
        auto do_step = [&](const int u, const uint4 cw, const uint4 cy, const uint32_t cb, const uint4 cb4, auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            const int par = u & 1;
            const int x = u - SKEW * r;
            const bool inside = !EDGE || (row_ok && x >= 0 && x < W);
            const int xc = EDGE ? min(max(x, 0), W - 1) : x;
            // the row above's states of the previous step (pixel x+1 there): the from-above-right predecessor now (4 paths:
            // the from-above one); the other two are for the coming steps and are read further down, when registers are free
            uint4 nNow, nLo, nHi;
            if constexpr (R16) { nLo = sR16[par ^ 1][0][r * LPP + j]; nHi = sR16[par ^ 1][1][r * LPP + j]; }
            else nNow = sSt[par ^ 1][P8 ? 2 : 0][r * LPP + j];
            uint32_t CP[8], Y[8], YS[8], S[8];
            unpack_cb(cw, CP, Bpk);
            const bool top = EDGE && y == 0;                   // row 0 of the frame: every path from above starts (:152-180)
            // from the left (-1,0): :183-191
            {
                const bool st = EDGE && x <= 0;
                if (st) {
#pragma unroll
                    for (int i = 0; i < 8; i++) FS[i] = P2pk;
                }
                step_b<LPP, EDGE>(FS, CP, YS, P1pk, P2, sel, st ? 0u : 0xFFFFu);
            }
            if (MODE == 2) {
                // the first pass's sum joins the running sum right away: its five registers are free for the rest of the step
                uint32_t E2[8];
                if (Y16) { E2[0] = cy.x; E2[1] = cy.y; E2[2] = cy.z; E2[3] = cy.w; E2[4] = cb4.x; E2[5] = cb4.y; E2[6] = cb4.z; E2[7] = cb4.w; }
                else if (BITS) unpack_p9(cy, cb, E2);
                else unpack_p(cy, E2);
#pragma unroll
                for (int i = 0; i < 8; i++) YS[i] += E2[i];
            }
            __builtin_amdgcn_sched_barrier(0);
            // from above (0,-1): :193-202
            uint4 newU, newD, newR;
            {
                unpack_p(P8 ? hU : nNow, S);
                if (top) {
#pragma unroll
                    for (int i = 0; i < 8; i++) S[i] = P2pk;
                }
                step_b<LPP, EDGE>(S, CP, Y, P1pk, P2, sel, top ? 0u : 0xFFFFu);
                newU = pack_p(S);
                sSt[par][0][(r + 1) * LPP + j] = newU;
#pragma unroll
                for (int i = 0; i < 8; i++) YS[i] += Y[i];
            }
            __builtin_amdgcn_sched_barrier(0);                 // one path after the other: interleaving them costs registers (spills)
            if constexpr (P8) {
                // from above-left (-1,-1): :205-213
                {
                    const bool st = top || (EDGE && x <= 0);
                    unpack_p(hD2, S);
                    if (st) {
#pragma unroll
                        for (int i = 0; i < 8; i++) S[i] = P2pk;
                    }
                    step_b<LPP, EDGE>(S, CP, Y, P1pk, P2, sel, st ? 0u : 0xFFFFu);
                    newD = pack_p(S);
                    sSt[par][1][(r + 1) * LPP + j] = newD;
#pragma unroll
                    for (int i = 0; i < 8; i++) YS[i] += Y[i];
                    hD2 = hD1;
                    hD1 = sSt[par ^ 1][1][r * LPP + j];          // pixel x+1's from-above-left state: used two steps on
                    hU = sSt[par ^ 1][0][r * LPP + j];           // its from-above state: used next step
                }
                __builtin_amdgcn_sched_barrier(0);
                // from above-right (+1,-1): :215-225
                {
                    const bool st = top || (EDGE && x >= W - 1);
                    if (R16) { S[0] = nLo.x; S[1] = nLo.y; S[2] = nLo.z; S[3] = nLo.w; S[4] = nHi.x; S[5] = nHi.y; S[6] = nHi.z; S[7] = nHi.w; }
                    else unpack_p(nNow, S);
                    if (st) {
#pragma unroll
                        for (int i = 0; i < 8; i++) S[i] = P2pk;
                    }
                    step_b<LPP, EDGE>(S, CP, Y, P1pk, P2, sel, st ? 0u : 0xFFFFu);
                    if constexpr (R16) {
                        sR16[par][0][(r + 1) * LPP + j] = make_uint4(S[0], S[1], S[2], S[3]);
                        sR16[par][1][(r + 1) * LPP + j] = make_uint4(S[4], S[5], S[6], S[7]);
                        if (!last_band && r == R - 1) newR = pack_p(S);      // only the band's last row hands it on as bytes
                    } else {
                        newR = pack_p(S);
                        sSt[par][2][(r + 1) * LPP + j] = newR;
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) YS[i] += Y[i];
                }
            }
            // the band's last row hands its states to the band below, by image column
            if (!last_band && r == R - 1 && inside) {
                uint8_t* o = Eout + ((uint32_t)xc * (NST * LPP) + (uint32_t)j) * 16u;
                if (CHAIN) {                                   // read while this band runs: tagged words, past the L1, coherent across the XCDs
                    edge_store((uint4*)o, make_uint4(newU.x | tag, newU.y | tag, newU.z | tag, newU.w | tag));
                    if constexpr (P8) {
                        edge_store((uint4*)(o + LPP * 16), make_uint4(newD.x | tag, newD.y | tag, newD.z | tag, newD.w | tag));
                        edge_store((uint4*)(o + 2 * LPP * 16), make_uint4(newR.x | tag, newR.y | tag, newR.z | tag, newR.w | tag));
                    }
                } else {
                    *(uint4*)o = newU;
                    if constexpr (P8) { *(uint4*)(o + LPP * 16) = newD; *(uint4*)(o + 2 * LPP * 16) = newR; }
                }
            }
            if (MODE != 2) {
                // the sum of this pass's y (:227-232): low bytes + 9th bits
                if (inside) {
                    const uint32_t px = (uint32_t)pix_of(xc, yc);
                    if (Y16) {
                        vol_store(Yf + px * D + (uint32_t)j * 16, make_uint4(YS[0], YS[1], YS[2], YS[3]));
                        vol_store(Bf + px * D + (uint32_t)j * 16, make_uint4(YS[4], YS[5], YS[6], YS[7]));
                    } else {
                        vol_store(Yf + px * D + (uint32_t)j * 16, pack_p(YS));
                        if (BITS) *(uint32_t*)(Bf + (px * LPP + (uint32_t)j) * 4u) = pack_hi_bits(YS);
                    }
                }
            } else {
                // S = PATHS*(C + P2) - (this pass's y + the first pass's), WTA on the spot (:227-232, :259-275)
                uint32_t ST[8];
#pragma unroll
                for (int i = 0; i < 8; i++) ST[i] = pk_mad16(CP[i], (uint32_t)PATHS * 0x10001u, 0u) - YS[i];    // the P1 biases of CP and YS cancel; no borrow between the halves
                if constexpr (RECLDS)
                    wta_row_record_at<LPP, NWV * 64, true, true>(ST, sRow, tid, j, inside, recb, s0b, 0u, &sRec[wave * 64 + g * 8 + (u & 7)], &sRs0[wave * 64 + g * 8 + (u & 7)]);
                else
                    wta_row_record_at<LPP, NWV * 64, true>(ST, sRow, tid, j, inside, recb, s0b, (uint32_t)pix_of(xc, yc));
                if (TAP && inside) {                           // debug tap (an instantiation of its own): S in natural d order
                    uint32_t* o = a.Sdbg + (f * (size_t)NP + pix_of(xc, yc)) * D + j * 16;
#pragma unroll
                    for (int i = 0; i < 8; i++) { o[i] = ST[i] & 0xFFFFu; o[i + 8] = ST[i] >> 16; }
                }
            }
        };
        // RECLDS: the records of steps u - 7 .. u of the wave's eight rows -> HBM, lane = (row, step): eight lanes of a row store
        // eight consecutive pixels (64 + 16 contiguous bytes).  A slot that holds no record (pixel outside the image, step not run)
        // carries the marker; every slot is reset behind its read
        auto flush_records = [&](const int u) {
            if constexpr (RECLDS) {
                __builtin_amdgcn_wave_barrier();
                const int fr = lane >> 3, fs = lane & 7;                              // row of the wave, slot
                const uint2 rc = sRec[wave * 64 + lane];
                const uint16_t s0v = sRs0[wave * 64 + lane];
                sRec[wave * 64 + lane] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                const int uu = (u & ~7) + fs, rr = wave * PXG + fr;
                const int xx = uu - SKEW * rr;
                if (rc.x != 0xFFFFFFFFu) {                                            // (a real record's fields are all below 2^16)
                    const uint32_t px = (uint32_t)pix_of(xx, yb + rr);
                    *(uint2*)(recb + px * 8u) = rc;
                    *(uint16_t*)(s0b + px * 2u) = s0v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        auto step = [&](const int u, const uint4 cw, const uint4 cy, const uint32_t cb, const uint4 cb4) {
            if (wave == 0 && have_above) {                                            // "row -1" of step u: column u + SKEW of the band above
                const uint4 v = settle(ringE[0], edge_at(u + SKEW), loader);
#pragma unroll
                for (int i = 0; i + 1 < PFE; i++) ringE[i] = ringE[i + 1];       // (wave 0 only)
                if (loader) {
                    put_above(u & 1, v);
                    ringE[PFE - 1] = eload(edge_at(u + PFE + SKEW));
                }
            }
            if (wave_rows && u >= act_lo && u <= act_hi) {                            // wave-uniform
                if (wave_plain_rows && u >= pl_lo && u <= pl_hi) do_step(u, cw, cy, cb, cb4, std::false_type{});
                else do_step(u, cw, cy, cb, cb4, std::true_type{});
            }
            if constexpr (RECLDS) {
                if ((u & 7) == 7) flush_records(u);                                   // workgroup-uniform
            }
#ifndef FSGM_BAND_NOBAR                                                  /* timing experiment only: wrong results without it */
            __syncthreads();                                   // states of step u visible to step u+1
#endif
        };

        int u0 = 0;
        for (; u0 + PF <= nsteps; u0 += PF) {
#pragma unroll
            for (int i = 0; i < PF; i++) {
                const int u = u0 + i;
                const uint4 cw = ringC[i];
                const uint4 cy = ringY[MODE == 2 ? i : 0];
                const uint32_t cb = ringB[MODE == 2 && BITS && !Y16 ? i : 0];
                const uint4 cb4 = ringB4[MODE == 2 && Y16 ? i : 0];
                const uint32_t off = vox_off(u + PF);
                ringC[i] = *(const uint4*)(Cf + off);
                if (MODE == 2) {
                    ringY[i] = vol_load(Yf + off);
                    if (Y16) ringB4[i] = vol_load(Bf + off);
                    else if (BITS) ringB[i] = *(const uint32_t*)(Bf + bit_off(u + PF));
                }
                step(u, cw, cy, cb, cb4);
            }
        }
#pragma unroll
        for (int i = 0; i < PF - 1; i++)
            if (u0 + i < nsteps) step(u0 + i, ringC[i], ringY[MODE == 2 ? i : 0], ringB[MODE == 2 && BITS && !Y16 ? i : 0], ringB4[MODE == 2 && Y16 ? i : 0]);   // workgroup-uniform
        if constexpr (RECLDS) {
            if ((nsteps & 7) != 0) flush_records(nsteps - 1);  // the last, partial group of steps
        }
        if (!CHAIN) {                                          // the band below reads what the last row stored: stores done before anyone goes on
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
size_t band_edge_uint4s(int W, int D, int paths) { const int lpp = agg_packed_lpp(D); return (size_t)W * (paths == 8 ? 3 : 1) * lpp; }   // per frame and band boundary
// per frame: the bit plane (one dword per lane) or, with FSGM_BAND_Y16, registers 4-7 of every lane (16 bytes: as large as C)
size_t band_bits_u32s(int W, int H, int D) { return (size_t)W * H * agg_packed_lpp(D) * (FSGM_BAND_Y16 ? 4 : 1); }
// the first pass's sum of four (y + P1) needs a 9th bit above 255
bool band_needs_bits(int paths, int P1, int P2) { return paths == 8 && 4 * (P1 + P2) > 255; }
// D = 16 << k; P1 + P2 <= 127: states and the biased y as bytes, two / four of them summed in 8 / 9 bits;
// 16 * paths * (cmax + P2 + P1) + 15 < 0x7C00: the WTA's packed keys stay below the fp16 infinity pattern (epi_step.h)
bool band_ok(int D, int paths, int P1, int P2, int cmax) {
    return agg_packed_lpp(D) != 0 && P1 >= 0 && P2 >= 0 && P1 + P2 <= 127 && (paths == 8 || paths == 4) && 16 * paths * (cmax + P2 + P1) + 15 < 0x7C00;
}

template <int LPP, int MODE>
static void launch_band_t(hipStream_t st, const BandArgs& a, int frames, int paths) {
    constexpr int NWV = FSGM_BAND_WAVES;
    const bool bits = band_needs_bits(paths, a.P1, a.P2);
    dim3 grid((unsigned)(a.chain ? frames * a.nbands : frames)), block(NWV * 64);
    if (MODE == 2 && a.Sdbg) {                               // the S debug tap: instantiations of their own, none of it in the product kernels
        if (paths == 4)  hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 4, false, MODE == 2, false>), grid, block, 0, st, a);
        else if (!bits)  hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 8, false, MODE == 2, false>), grid, block, 0, st, a);
        else             hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 8, true, MODE == 2, false>), grid, block, 0, st, a);
        return;
    }
    if (a.chain) {
        if (paths == 4)  hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 4, false, false, true>), grid, block, 0, st, a);
        else if (!bits)  hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 8, false, false, true>), grid, block, 0, st, a);
        else             hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 8, true, false, true>), grid, block, 0, st, a);
        return;
    }
    if (paths == 4)  hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 4, false, false, false>), grid, block, 0, st, a);
    else if (!bits)  hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 8, false, false, false>), grid, block, 0, st, a);
    else             hipLaunchKernelGGL((band_kernel<LPP, MODE, NWV, 8, true, false, false>), grid, block, 0, st, a);
}

int band_rows(int D) { const int lpp = agg_packed_lpp(D); return lpp ? FSGM_BAND_WAVES * (64 / lpp) : 0; }   // rows per band

// One whole pass of `frames` frames: mode 0 = first pass -> Y (+ bit plane), mode 2 = second pass + WTA records
void launch_band(hipStream_t st, const BandArgs& a, int frames, int paths, int mode) {
#define FSGM_BAND(L) do { if (mode == 0) launch_band_t<L, 0>(st, a, frames, paths); else launch_band_t<L, 2>(st, a, frames, paths); } while (0)
    switch (agg_packed_lpp(a.D)) {
        case 1: FSGM_BAND(1); break;
        case 2: FSGM_BAND(2); break;
        case 4: FSGM_BAND(4); break;
        case 8: FSGM_BAND(8); break;
        case 16: FSGM_BAND(16); break;
        default: break;
    }
#undef FSGM_BAND
}

}  // namespace fsgm
