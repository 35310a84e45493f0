// epi_sweep.hip -- fused-sweep aggregation for the 8-path calc_cost_sgm case (gfx950).
//
// Why: the per-direction kernel (epi_kernels.hip, agg_packed_kernel) moves 24 B per voxel through
// HBM (8 reads of C, 8 writes + 8 reads of L_r) and is bandwidth-bound at ~1/5 of the algorithmic
// roofline.  Here the three paths that advance row by row -- from above (0,+1), from above-left
// (+1,+1) and from above-right (-1,+1), calc_cost_sgm.cpp:193-226 -- are computed TOGETHER for the
// same pixel, so C is read once per sweep and only their sum leaves the chip:
//
//   pair kernels (AXIS 0)   C -> X_h  = (L_left - C) + (L_right - C), the two horizontal paths      (u8)
//                           (checkpoint-and-recompute, see the pair section below)
//   down sweep   (MODE 0)   C -> X_dn = sum over the three pass-0 paths from above of (L_r - C)  (u8)
//   final sweep  (MODE 2)   on the point-mirrored frame (pass 1): its own three paths, then in
//                           registers S = X_up + X_dn + X_h + 8*C and the WTA;
//                           writes one 18-byte record per PIXEL
//   sweep_finish_kernel     parabola / vz->disparity from the records -> bestD, minC
//
// Every path cost satisfies C <= L_r <= C + P2 when nothing wraps, so the EXCESS L_r - C of three
// paths fits a byte whenever 3*P2 <= 255 (the reference uses P2 = 64 and 32): X_dn and X_h cost
// 1 B per voxel each, and neither X_up nor S (u32 in the reference) ever reaches HBM.  8.25 B per
// voxel by design (9.65 measured) instead of 24.  One sweep launch alone -- strips x frames
// workgroups -- is too small to fill 256 CUs, so the host forks the work over three streams: the
// pair kernels, and two lanes of frames that each sweep down and then up (capi_epi.hip).
// (lh_planes = 2 keeps the earlier form: both horizontal path volumes from agg_packed_kernel,
// S = ... + 6*C + L_left + L_right; FSGM_EPI_HPAIR=0, for A/B runs.)
// MODE 1 (plain up sweep writing X_up) and wta_sweep_kernel exist for the debug tap that
// rebuilds S in natural order (fsgm_epi_plan_download_sum).
//
// The diagonal paths couple neighbouring columns, so a workgroup that owns a strip of columns needs
// its neighbours' boundary values every row.  Instead of in-kernel neighbour synchronisation the
// sweep is cut into blocks of T rows (one launch per block) and every workgroup recomputes a halo
// of T columns on each side (only the one diagonal direction that flows inwards): a value that is
// wrong because ITS predecessor lay outside the halo moves one column per row and therefore cannot
// reach the strip within T rows.  At block boundaries the three path states of every column go
// through a small global buffer.  No spin-waits, no co-residency assumption.
//
// Workgroup = 4 waves; wave w owns PXW = 64/LPP adjacent columns (same lane layout as
// agg_packed_kernel: LPP lanes x 16 d per pixel); strip = 4*PXW columns; T = 2*PXW, so the halo is
// exactly 2 pixel groups per side = one extra unit of work per wave per row (balanced).
// Per row every wave runs 4 DP steps: vertical / both diagonals for its own columns + 1 halo step.
// The diagonal states shift one column per row through LDS (double buffered, one barrier per row).
//
// Path states are kept NORMALISED: L' = L - m, with m the stored minimum of that pixel (0 at a path
// start, calc_cost_sgm.cpp:154).  Then  L_new = C + min( min(L'[d], min(L'[d-1],L'[d+1]) + P1), P2 )
// needs no separate m.  Valid under the same no-wrap precondition as agg_packed_kernel<.,false>
// (max C + P2 + max(P1,P2) <= 255, P1,P2 >= 0); other parameters use the per-direction kernels.
#include "epi_kernels.h"
#include "fsgm_device.h"
#include "epi_wta_tail.h"

namespace fsgm {

namespace {

constexpr uint32_t SEL_E = 0x0C020C00u;     // v_perm: bytes 0,2 -> 2 x u16
constexpr uint32_t SEL_O = 0x0C030C01u;     // v_perm: bytes 1,3 -> 2 x u16
constexpr uint32_t SEL_PACK = 0x06020400u;  // v_perm(O, E): bytes E.lo, O.lo, E.hi, O.hi

__device__ __forceinline__ uint32_t pk_mad16(uint32_t a, uint32_t b, uint32_t c) {        // v_pk_mad_u16
    u16x2 r = __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void unpack16(const uint4 v, uint32_t (&E)[4], uint32_t (&O)[4]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        E[k] = __builtin_amdgcn_perm(0u, w[k], SEL_E);
        O[k] = __builtin_amdgcn_perm(0u, w[k], SEL_O);
    }
}
__device__ __forceinline__ uint4 pack16(const uint32_t (&E)[4], const uint32_t (&O)[4]) {
    uint4 o;
    o.x = __builtin_amdgcn_perm(O[0], E[0], SEL_PACK);
    o.y = __builtin_amdgcn_perm(O[1], E[1], SEL_PACK);
    o.z = __builtin_amdgcn_perm(O[2], E[2], SEL_PACK);
    o.w = __builtin_amdgcn_perm(O[3], E[3], SEL_PACK);
    return o;
}

// One DP step on normalised state (calc_cost_sgm.cpp:33-66).  LE/LO: previous pixel's L - m;
// on return they hold the new pixel's normalised state, and XE/XO its excess L_new - C
// (= min(L'[d], min(L'[d-1], L'[d+1]) + P1, P2), in [0, P2]).
template <int LPP>
__device__ __forceinline__ void step_norm(uint32_t (&LE)[4], uint32_t (&LO)[4], const uint32_t (&CE)[4],
                                          const uint32_t (&CO)[4], uint32_t (&XE)[4], uint32_t (&XO)[4],
                                          const bool start, const uint32_t P1pk, const uint32_t P2pk, const int j) {
    uint32_t NE[4], NO[4];
    constexpr uint32_t SENT = 0xFFFFFFFFu;
    uint32_t prevO3 = dpp_mov<DPP_ROW_SHR1>(SENT, LO[3]);
    uint32_t nextE0 = dpp_mov<DPP_ROW_SHL1>(SENT, LE[0]);
    if (j == 0) prevO3 = SENT;                       // d = 0 has no d-1     (:47)
    if (j == LPP - 1) nextE0 = SENT;                 // d = D-1 has no d+1   (:48)
    const uint32_t p2lane = start ? 0u : P2pk;       // min(., 0) = 0  ->  L = C at a path start (:152-180)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t nbE = pk_min(align16(LO[k], k ? LO[k - 1] : prevO3), LO[k]);
        const uint32_t nbO = pk_min(LE[k], align16(k < 3 ? LE[k + 1] : nextE0, LE[k]));
        XE[k] = pk_min(pk_min(LE[k], pk_add(nbE, P1pk)), p2lane);
        XO[k] = pk_min(pk_min(LO[k], pk_add(nbO, P1pk)), p2lane);
        NE[k] = pk_add(CE[k], XE[k]);
        NO[k] = pk_add(CO[k], XO[k]);
    }
    const uint32_t mm = pk_min(pk_min(pk_min(NE[0], NO[0]), pk_min(NE[1], NO[1])),
                               pk_min(pk_min(NE[2], NO[2]), pk_min(NE[3], NO[3])));
    uint32_t mx = min(mm & 0xFFFFu, mm >> 16);
    mx = group_min_u32<LPP>(mx);
    mx = start ? 0u : mx;                            // stored minimum 0 at a path start (:154,:164)
    const uint32_t mpk = mx | (mx << 16);
#pragma unroll
    for (int k = 0; k < 4; k++) { LE[k] = pk_sub(NE[k], mpk); LO[k] = pk_sub(NO[k], mpk); }
}

// Per-pixel WTA of the final passes: S (packed u16, E/O split) of the LPP lanes of a pixel -> one record
// {best, minC, S[best-1], S[best+1]} + S[0] (calc_cost_sgm.cpp:263-271; the parabola runs in
// sweep_finish_kernel).  First minimum over d: inside a lane as packed u16 keys S*16 + (index in the
// lane) (S <= 8*255, so a key fits 16 bits and two of them compare per v_pk_min_u16); across the lanes
// of a pixel as (S << 8 | d).  sRow: 8 u32 per lane, rows private to the wave that writes them.
template <int LPP>
__device__ __forceinline__ void wta_row_record(const uint32_t (&SE)[4], const uint32_t (&SO)[4], uint32_t* sRow, int tid, int j,
                                               bool ok, uint4* rec, uint16_t* s0, size_t idx) {
    constexpr int D = LPP * 16;
    uint32_t* row = sRow + (size_t)(tid / LPP) * (D / 2) + j * 8;
    uint32_t kmin = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        row[2 * q] = __builtin_amdgcn_perm(SO[q], SE[q], 0x05040100u);       // (S[4q], S[4q+1])   natural d order for the
        row[2 * q + 1] = __builtin_amdgcn_perm(SO[q], SE[q], 0x07060302u);   // (S[4q+2], S[4q+3]) parabola taps
        const uint32_t kE = pk_mad16(SE[q], 0x00100010u, (uint32_t)(4 * q) | ((uint32_t)(4 * q + 2) << 16));
        const uint32_t kO = pk_mad16(SO[q], 0x00100010u, (uint32_t)(4 * q + 1) | ((uint32_t)(4 * q + 3) << 16));
        kmin = pk_min(kmin, pk_min(kE, kO));
    }
    const uint32_t k16 = min(kmin & 0xFFFFu, kmin >> 16);
    uint32_t key = ((k16 >> 4) << 8) | ((uint32_t)j * 16u + (k16 & 15u));
    key = group_min_u32<LPP>(key);
    __builtin_amdgcn_wave_barrier();
    if (j == 0 && ok) {
        const uint32_t best = key & 0xFF, minc = key >> 8;
        const uint16_t* srow = (const uint16_t*)(sRow + (size_t)(tid / LPP) * (D / 2));
        const uint32_t c_1 = best > 0 ? srow[best - 1] : 0u;
        const uint32_t c1 = best + 1 < (uint32_t)D ? srow[best + 1] : 0u;   // best == D-1: the finish kernel takes the next pixel's S[0]
        rec[idx] = make_uint4(best, minc, c_1, c1);
        s0[idx] = (uint16_t)srow[0];
    }
}

}  // namespace

// =============================================================================================
// sweep kernel: rows [y0, y0+rows) of the sweep frame.
//   MODE 0: pass-0 frame, writes X_dn.   MODE 1: point-mirrored frame, writes X_up.
//   MODE 2: point-mirrored frame, final: S = X_dn + X_up + 6C + L_left + L_right in registers,
//           WTA per pixel, writes one record {best, minC, S[best-1], S[best+1]} + S[0] per pixel.
// =============================================================================================
template <int LPP, int MODE, int NWV>
__global__ __launch_bounds__(NWV * 64) void sweep_kernel(SweepArgs a) {
    constexpr bool UP = MODE != 0;
    constexpr int PXW = 64 / LPP;            // columns per wave
    constexpr int D = LPP * 16;
    constexpr int STRIP = NWV * PXW;         // own columns per workgroup
    constexpr int T = (NWV / 2) * PXW;       // halo width = max rows per launch (one halo step per wave per row)
    constexpr int NCOL = STRIP + 2 * T + 2;  // LDS columns: forward column x  <->  index x - (a0 - T - 1)
    constexpr int PF = MODE == 2 ? 2 : 3;    // rows of C in flight per lane
    __shared__ uint4 sDiag[2][2][NCOL * LPP];    // [row parity][0: from above-left, 1: from above-right][column][lane-of-pixel]
    __shared__ uint32_t sRow[MODE == 2 ? NWV * 64 * 8 : 1];   // MODE 2: S of the wave's pixels in natural d order (u16)

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
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2pk = (uint32_t)a.P2 * 0x10001u;
    const int y0 = a.y0, rows = min(a.rows, H - y0);
    const bool first_block = y0 == 0;

    // columns in the (possibly mirrored) sweep frame
    const int gx = a0 + wave * PXW + g;                          // own column
    const bool own_ok = gx < W;
    const int gxc = min(gx, W - 1);
    // halo unit of this wave: waves 0,1 -> from-above-left on the T columns left of the strip,
    //                         waves 2,3 -> from-above-right on the T columns right of it
    const int hdir = wave >= NWV / 2 ? 1 : 0;
    const int hx = hdir == 0 ? a0 - T + wave * PXW + g : a0 + STRIP + (wave - NWV / 2) * PXW + g;
    const int hxc = min(max(hx, 0), W - 1);
    const int lbase = a0 - T - 1;                                // forward column of LDS index 0

    auto pix_of = [&](int x, int y) -> int {                     // actual pixel index of sweep-frame (x,y)
        const int p = y * W + x;
        return UP ? NP - 1 - p : p;
    };
    auto vox_off = [&](int x, int y) -> uint32_t {               // byte offset of (x,y)'s 16 costs of this lane
        return (uint32_t)pix_of(x, y) * D + (uint32_t)j * 16;
    };

    // ---- block prologue: path states of the row above (from the previous launch) ----
    uint32_t VE[4] = {0, 0, 0, 0}, VO[4] = {0, 0, 0, 0};          // vertical path state of the own column
    if (!first_block) {
        unpack16(*(const uint4*)(StIn + (size_t)gxc * D + j * 16), VE, VO);
        for (int i = tid; i < 2 * NCOL * LPP; i += NWV * 64) {
            const int dir = i / (NCOL * LPP), r = i - dir * (NCOL * LPP);
            const int c = r / LPP, jj = r - c * LPP;
            const int x = min(max(lbase + c, 0), W - 1);
            sDiag[0][dir][r] = *(const uint4*)(StIn + ((size_t)(1 + dir) * W + x) * D + jj * 16);
        }
    }
    __syncthreads();

    uint4 ringOwn[PF], ringHalo[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) {
        const int y = min(y0 + i, H - 1);
        ringOwn[i] = *(const uint4*)(Cf + vox_off(gxc, y));
        ringHalo[i] = *(const uint4*)(Cf + vox_off(hxc, y));
    }

    const uint8_t* __restrict__ Lhf = MODE == 2 ? a.Lh + f * a.lh_frame_stride : nullptr;

    // one row of the block: 4 DP steps per wave, one barrier
    auto do_row = [&](const int k, const uint4 cOwn, const uint4 cHalo) {
        const int y = y0 + k, par = k & 1;
        const bool top = y == 0;
        // MODE 2: the other partial sums of the own pixel; issued now, consumed after the four DP
        // steps of this row, which hide most of their latency
        uint4 curX, curL0, curL1;
        if (MODE == 2) {
            const uint32_t off = vox_off(gxc, y);
            curX = *(const uint4*)(Xf + off);
            curL0 = *(const uint4*)(Lhf + off);
            if (a.lh_planes == 2) curL1 = *(const uint4*)(Lhf + a.lh_dir_stride + off);
        }
        uint32_t CE[4], CO[4], XE[4], XO[4], SE[4], SO[4];
        unpack16(cOwn, CE, CO);

        // from above (0,+1)                                            calc_cost_sgm.cpp:193-202
        step_norm<LPP>(VE, VO, CE, CO, XE, XO, top, P1pk, P2pk, j);
#pragma unroll
        for (int q = 0; q < 4; q++) { SE[q] = XE[q]; SO[q] = XO[q]; }

        // from above-left (+1,+1): predecessor column gx-1                        :205-213
        {
            uint32_t LE[4], LO[4];
            unpack16(sDiag[par][0][(gx - 1 - lbase) * LPP + j], LE, LO);
            step_norm<LPP>(LE, LO, CE, CO, XE, XO, top || gx == 0, P1pk, P2pk, j);
            sDiag[par ^ 1][0][(gx - lbase) * LPP + j] = pack16(LE, LO);
#pragma unroll
            for (int q = 0; q < 4; q++) { SE[q] += XE[q]; SO[q] += XO[q]; }
        }
        // from above-right (-1,+1): predecessor column gx+1                       :215-225
        {
            uint32_t LE[4], LO[4];
            unpack16(sDiag[par][1][(gx + 1 - lbase) * LPP + j], LE, LO);
            step_norm<LPP>(LE, LO, CE, CO, XE, XO, top || gx == W - 1, P1pk, P2pk, j);
            sDiag[par ^ 1][1][(gx - lbase) * LPP + j] = pack16(LE, LO);
#pragma unroll
            for (int q = 0; q < 4; q++) { SE[q] += XE[q]; SO[q] += XO[q]; }
        }
        // halo unit: keeps the inward-flowing diagonal correct for the next rows
        {
            uint32_t HE[4], HO[4], LE[4], LO[4];
            unpack16(cHalo, HE, HO);
            const int px = hdir == 0 ? hx - 1 : hx + 1;
            unpack16(sDiag[par][hdir][(px - lbase) * LPP + j], LE, LO);
            const bool st = top || (hdir == 0 ? hx == 0 : hx == W - 1);
            step_norm<LPP>(LE, LO, HE, HO, XE, XO, st, P1pk, P2pk, j);
            sDiag[par ^ 1][hdir][(hx - lbase) * LPP + j] = pack16(LE, LO);
        }
        if (MODE != 2) {
            // excess sum of this sweep's three paths, one byte per voxel (3*P2 <= 255)      :227-232
            if (own_ok) *(uint4*)(Xf + vox_off(gx, y)) = pack16(SE, SO);
        } else {
            // S = X_up (registers) + X_dn + 6*C + from-the-left + from-the-right, all at this pixel
            // (lh_planes = 1: the horizontal pair arrives as its excess sum X_h, so 8*C)
            uint32_t E2[4], O2[4];
            const uint32_t nC = a.lh_planes == 2 ? 6u : 8u;
            unpack16(curX, E2, O2);                                               // X_dn
#pragma unroll
            for (int q = 0; q < 4; q++) { SE[q] += E2[q] + nC * CE[q]; SO[q] += O2[q] + nC * CO[q]; }
            unpack16(curL0, E2, O2);
#pragma unroll
            for (int q = 0; q < 4; q++) { SE[q] += E2[q]; SO[q] += O2[q]; }
            if (a.lh_planes == 2) {
                unpack16(curL1, E2, O2);
#pragma unroll
                for (int q = 0; q < 4; q++) { SE[q] += E2[q]; SO[q] += O2[q]; }
            }
            wta_row_record<LPP>(SE, SO, sRow, tid, j, own_ok, a.rec, a.s0, f * (size_t)NP + pix_of(gx, y));
        }
        __syncthreads();                                         // diagonal states of row y visible to row y+1
    };

    // steady state without branches so the prefetched rows stay in flight; then the tail
    int k0 = 0;
    for (; k0 + PF <= rows; k0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const uint4 cOwn = ringOwn[i], cHalo = ringHalo[i];
            const int yn = min(y0 + k0 + i + PF, H - 1);
            ringOwn[i] = *(const uint4*)(Cf + vox_off(gxc, yn));
            ringHalo[i] = *(const uint4*)(Cf + vox_off(hxc, yn));
            do_row(k0 + i, cOwn, cHalo);
        }
    }
#pragma unroll
    for (int i = 0; i < PF - 1; i++)
        if (k0 + i < rows) do_row(k0 + i, ringOwn[i], ringHalo[i]);   // block-uniform

    // ---- block epilogue: states of the last row for the next launch ----
    if (y0 + rows < H && own_ok) {
        const int par = rows & 1;                                // buffer the last row wrote into
        *(uint4*)(StOut + (size_t)gx * D + j * 16) = pack16(VE, VO);
        *(uint4*)(StOut + ((size_t)W + gx) * D + j * 16) = sDiag[par][0][(gx - lbase) * LPP + j];
        *(uint4*)(StOut + ((size_t)2 * W + gx) * D + j * 16) = sDiag[par][1][(gx - lbase) * LPP + j];
    }
}

// =============================================================================================
// finish kernel for MODE 2: parabola + vz->disparity from the per-pixel records
// (calc_cost_sgm.cpp:278-308, :414-426).  best == D-1 reads the next pixel's S[0] (:296).
// =============================================================================================
__global__ __launch_bounds__(256) void sweep_finish_kernel(WtaArgs a, const uint4* __restrict__ rec,
                                                           const uint16_t* __restrict__ s0) {
    const int NP = a.W * a.H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const size_t f = blockIdx.y;
    const uint4 r = rec[f * (size_t)NP + p];
    uint32_t c1 = r.w;
    if (r.x + 1 == (uint32_t)a.D) c1 = p + 1 < NP ? (uint32_t)s0[f * (size_t)NP + p + 1] : 0u;
    wta_finish(a, f, p, r.x, r.y, r.z, c1);
}

// =============================================================================================
// WTA over S = X_dn + X_up + 6*C + L_left + L_right  (calc_cost_sgm.cpp:227-232, :259-308, :414-426)
// for the non-final mode (X_up materialised by a MODE 1 sweep).
// =============================================================================================
template <int LPP>
__global__ __launch_bounds__(256) void wta_sweep_kernel(WtaArgs a, SweepSumArgs q) {
    constexpr int D = LPP * 16;
    constexpr int PPB = 256 / LPP;
    __shared__ uint32_t sS[256 * 8];
    const int tid = threadIdx.x;
    const int NP = a.W * a.H;
    const int gp = blockIdx.x * PPB + tid / LPP, j = tid % LPP;
    const bool valid = gp < NP;
    const int p = valid ? gp : NP - 1;
    const size_t f = blockIdx.y;
    const size_t bo = (size_t)p * D + (size_t)j * 16;                    // byte offset in a u8 volume
    const uint8_t* Lh = q.Lh + f * q.lh_frame_stride;
    uint32_t E[4], O[4], E2[4], O2[4];
    unpack16(*(const uint4*)(q.C + f * q.v_frame_stride + bo), E, O);
    const uint32_t nC = (uint32_t)q.nC;
#pragma unroll
    for (int k = 0; k < 4; k++) { E[k] *= nC; O[k] *= nC; }
    unpack16(*(const uint4*)(q.Xdn + f * q.v_frame_stride + bo), E2, O2);
#pragma unroll
    for (int k = 0; k < 4; k++) { E[k] += E2[k]; O[k] += O2[k]; }
    if (q.Xup) {
        unpack16(*(const uint4*)(q.Xup + f * q.v_frame_stride + bo), E2, O2);
#pragma unroll
        for (int k = 0; k < 4; k++) { E[k] += E2[k]; O[k] += O2[k]; }
    }
    unpack16(*(const uint4*)(Lh + bo), E2, O2);
#pragma unroll
    for (int k = 0; k < 4; k++) { E[k] += E2[k]; O[k] += O2[k]; }
    if (q.lh_planes == 2) {
        unpack16(*(const uint4*)(Lh + q.lh_dir_stride + bo), E2, O2);
#pragma unroll
        for (int k = 0; k < 4; k++) { E[k] += E2[k]; O[k] += O2[k]; }
    }

    uint32_t key = 0xFFFFFFFFu;
    uint32_t* row = sS + (size_t)(tid / LPP) * (D / 2) + j * 8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t v0 = E[k] & 0xFFFF, v1 = O[k] & 0xFFFF, v2 = E[k] >> 16, v3 = O[k] >> 16;
        row[2 * k] = v0 | (v1 << 16);
        row[2 * k + 1] = v2 | (v3 << 16);
        const uint32_t dd = (uint32_t)j * 16 + 4 * k;
        key = min(key, min(min((v0 << 8) | dd, (v1 << 8) | (dd + 1)), min((v2 << 8) | (dd + 2), (v3 << 8) | (dd + 3))));
        if (q.Sdbg && valid) {
            uint32_t* o = q.Sdbg + f * (size_t)NP * D + (size_t)p * D + dd;
            o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3;
        }
    }
    key = group_min_u32<LPP>(key);
    __syncthreads();
    if (j == 0 && valid) {
        const uint32_t best = key & 0xFF, minc = key >> 8;
        const uint16_t* srow = (const uint16_t*)(sS + (size_t)(tid / LPP) * (D / 2));
        uint32_t c_1 = 0, c1 = 0;
        if (a.subpixel && best > 1) {
            c_1 = srow[best - 1];
            if (best + 1 < (uint32_t)D) c1 = srow[best + 1];
            else if (p + 1 < NP) {                                           // next pixel's d=0 (:296)
                const size_t nb = f * q.v_frame_stride + (size_t)(p + 1) * D;
                c1 = nC * q.C[nb] + q.Xdn[nb] + (q.Xup ? (uint32_t)q.Xup[nb] : 0u) + Lh[(size_t)(p + 1) * D] +
                     (q.lh_planes == 2 ? (uint32_t)Lh[q.lh_dir_stride + (size_t)(p + 1) * D] : 0u);
            }
        }
        wta_finish(a, f, p, best, minc, c_1, c1);
    }
}

// =============================================================================================
// An opposite pair of paths as ONE excess sum  X = (L_fwd - C) + (L_bwd - C)  (<= 2*P2, one byte).
// The two paths of an axis (calc_cost_sgm.cpp:183-202 and their pass-1 mirrors) run in opposite
// directions along a line, so their values for a pixel exist at different times; writing both path
// volumes and reading them back costs 6 B per voxel (C twice, 2 writes, 2 reads).
// Checkpoint-and-recompute brings that to ~4.25 B:
//   pass A (pair_ckpt_kernel)  end -> start of the line, keeps nothing but the normalised backward
//          state at every HP_TC-th position (1/HP_TC B per voxel);
//   pass B (pair_sum_kernel)   start -> end in tiles of HP_TC positions: the tile's backward excesses
//          are recomputed from the checkpoint on its far edge into registers, then the forward path
//          crosses the tile, adds them and stores X.  The tile's C stays in registers between the two.
// AXIS 0: lines = image rows (the horizontal pair; 64/LPP rows per wave, consecutive positions 16*LPP
//         bytes apart); AXIS 1: lines = image columns (the vertical pair; 64/LPP adjacent columns per
//         wave -- one contiguous run per step -- consecutive positions a row apart).
// FINAL:  instead of storing X, pass B adds the other axis' X and nC*C and does the WTA on the spot --
//         the 4-path pipeline (the reference's shipped configuration): horizontal pair -> X_h, vertical
//         pair final: S = X_v + X_h + 4*C, 7.5 B per voxel for 4 voxel-paths, S never in HBM.
// Same lane layout as agg_packed_kernel: LPP lanes x 16 d per pixel.
// =============================================================================================
#ifndef FSGM_HP_TC
#define FSGM_HP_TC 8            // tile width = checkpoint spacing in positions (A/B knob)
#endif
constexpr int HP_TC = FSGM_HP_TC;

template <int LPP, int AXIS>
__global__ __launch_bounds__(256) void pair_ckpt_kernel(PairArgs a) {
    constexpr int PXW = 64 / LPP, D = LPP * 16, PF = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPP, j = lane % LPP;
    const int nl = AXIS ? a.W : a.H, len = AXIS ? a.H : a.W;
    const int lg = (int)blockIdx.x * 4 + wave;
    if (lg * PXW >= nl) return;                                 // wave-uniform
    const int l = min(lg * PXW + g, nl - 1);                    // lines past the last redo the last (same bytes, same addresses)
    const int NT = (len + HP_TC - 1) / HP_TC;
    if (NT < 2) return;                                         // a single tile starts at the border: no checkpoint
    const size_t f = blockIdx.y;
    const size_t lstride = AXIS ? (size_t)D : (size_t)a.W * D, tstride = AXIS ? (size_t)a.W * D : (size_t)D;
    const uint8_t* __restrict__ Cl = a.C + f * a.c_frame_stride + (size_t)l * lstride + (size_t)j * 16;
    uint8_t* __restrict__ Kl = a.ckpt + f * a.ckpt_frame_stride + ((size_t)l * (NT - 1)) * D + (size_t)j * 16;
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2pk = (uint32_t)a.P2 * 0x10001u;
    uint32_t LE[4] = {0, 0, 0, 0}, LO[4] = {0, 0, 0, 0};
    auto load_c = [&](int t) -> uint4 { return *(const uint4*)(Cl + (size_t)max(t, 0) * tstride); };
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
            uint32_t CE[4], CO[4], XE[4], XO[4];
            unpack16(cw, CE, CO);
            step_norm<LPP>(LE, LO, CE, CO, XE, XO, t == len - 1, P1pk, P2pk, j);
            // positions below `last` in the final group are computed but not needed; t >= 0 always holds there
            if (t >= last && (t % HP_TC) == 0) *(uint4*)(Kl + (size_t)(t / HP_TC - 1) * D) = pack16(LE, LO);
        }
    }
}

template <int LPP, int AXIS, bool FINAL>
__global__ __launch_bounds__(256) void pair_sum_kernel(PairArgs a) {
    constexpr int PXW = 64 / LPP, D = LPP * 16, TC = HP_TC;
    __shared__ uint32_t sRow[FINAL ? 4 * 64 * 8 : 1];            // FINAL: S of the wave's pixels in natural d order (u16)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane / LPP, j = lane % LPP;
    const int W = a.W, NP = a.W * a.H;
    const int nl = AXIS ? a.W : a.H, len = AXIS ? a.H : a.W;
    const int lg = (int)blockIdx.x * 4 + wave;
    if (lg * PXW >= nl) return;
    const bool own_ok = lg * PXW + g < nl;
    const int l = min(lg * PXW + g, nl - 1);
    const int NT = (len + TC - 1) / TC;
    const size_t f = blockIdx.y;
    const size_t lstride = AXIS ? (size_t)D : (size_t)W * D, tstride = AXIS ? (size_t)W * D : (size_t)D;
    const size_t lbase = (size_t)l * lstride + (size_t)j * 16;
    const uint8_t* __restrict__ Cl = a.C + f * a.c_frame_stride + lbase;
    uint8_t* __restrict__ Xl = FINAL ? nullptr : a.X + f * a.x_frame_stride + lbase;     // !FINAL: this pair's excess sum, out
    const uint8_t* __restrict__ Ol = FINAL ? a.Xother + f * a.xo_frame_stride + lbase : nullptr;   // FINAL: the other pair's, in
    const uint8_t* __restrict__ Kl = a.ckpt + f * a.ckpt_frame_stride + ((size_t)l * max(NT - 1, 1)) * D + (size_t)j * 16;
    const uint32_t P1pk = (uint32_t)a.P1 * 0x10001u, P2pk = (uint32_t)a.P2 * 0x10001u;
    uint32_t LE[4] = {0, 0, 0, 0}, LO[4] = {0, 0, 0, 0};          // forward state, carried across tiles
    auto load_c = [&](int t) -> uint4 { return *(const uint4*)(Cl + (size_t)min(t, len - 1) * tstride); };
    auto load_k = [&](int t) -> uint4 { return *(const uint4*)(Kl + (size_t)min(t, max(NT - 2, 0)) * D); };
    uint4 cT[TC], cN[TC], kT = load_k(0), kN;
#pragma unroll
    for (int c = 0; c < TC; c++) cT[c] = load_c(c);
    for (int t = 0; t < NT; t++) {
        const int tb = t * TC;
#pragma unroll
        for (int c = 0; c < TC; c++) cN[c] = load_c(tb + TC + c);      // next tile, in flight while this one computes
        kN = load_k(t + 1);
        // backward path through the tile: positions past the line end come first and are wiped by the path
        // start at the last position (:152-180); a tile inside the line starts from its checkpoint
        uint32_t RE[4], RO[4];
        unpack16(kT, RE, RO);
        uint4 exR[TC];
#pragma unroll
        for (int c = TC - 1; c >= 0; c--) {
            const int x = tb + c;
            uint32_t CE[4], CO[4], XE[4], XO[4];
            unpack16(cT[c], CE, CO);
            step_norm<LPP>(RE, RO, CE, CO, XE, XO, x >= len - 1, P1pk, P2pk, j);
            exR[c] = pack16(XE, XO);
        }
        uint4 xo[TC];
        if (FINAL) {
#pragma unroll
            for (int c = 0; c < TC; c++) xo[c] = *(const uint4*)(Ol + (size_t)min(tb + c, len - 1) * tstride);
        }
        // forward path, adding the two excesses
#pragma unroll
        for (int c = 0; c < TC; c++) {
            const int x = tb + c;
            uint32_t CE[4], CO[4], XE[4], XO[4];
            unpack16(cT[c], CE, CO);
            step_norm<LPP>(LE, LO, CE, CO, XE, XO, x == 0, P1pk, P2pk, j);
            // both excesses are <= P2 per byte and 2*P2 <= 255: the packed bytes add as plain words
            uint4 xs = pack16(XE, XO);
            xs.x += exR[c].x; xs.y += exR[c].y; xs.z += exR[c].z; xs.w += exR[c].w;
            if (!FINAL) {
                if (x < len) *(uint4*)(Xl + (size_t)x * tstride) = xs;
            } else {
                // S = this pair + the other pair + nC*C (calc_cost_sgm.cpp:227-232), WTA on the spot
                uint32_t SE[4], SO[4], E2[4], O2[4];
                unpack16(xs, SE, SO);
                unpack16(xo[c], E2, O2);
                const uint32_t nC = (uint32_t)a.nC;
#pragma unroll
                for (int q = 0; q < 4; q++) { SE[q] += E2[q] + nC * CE[q]; SO[q] += O2[q] + nC * CO[q]; }
                const int ap = AXIS ? x * W + l : l * W + x;     // pixel index (wave-uniform validity: x < len)
                wta_row_record<LPP>(SE, SO, sRow, tid, j, own_ok && x < len, a.rec, a.s0, f * (size_t)NP + (size_t)min(ap, NP - 1));
            }
        }
#pragma unroll
        for (int c = 0; c < TC; c++) cT[c] = cN[c];
        kT = kN;
    }
}

size_t pair_ckpt_bytes(int W, int H, int D, int axis) {
    const int len = axis ? H : W, nl = axis ? W : H;
    const int NT = (len + HP_TC - 1) / HP_TC;
    return (size_t)nl * (size_t)(NT > 1 ? NT - 1 : 1) * D;
}

template <int LPP>
static void launch_pair_t(hipStream_t st, const PairArgs& a, int frames, int axis, bool final_pass, int phase) {
    constexpr int PXW = 64 / LPP;
    const int nl = axis ? a.W : a.H;
    dim3 grid((nl + 4 * PXW - 1) / (4 * PXW), frames);
    if (phase != 2) {
        if (axis == 0) hipLaunchKernelGGL((pair_ckpt_kernel<LPP, 0>), grid, dim3(256), 0, st, a);
        else           hipLaunchKernelGGL((pair_ckpt_kernel<LPP, 1>), grid, dim3(256), 0, st, a);
    }
    if (phase != 1) {
        if (axis == 0 && !final_pass)      hipLaunchKernelGGL((pair_sum_kernel<LPP, 0, false>), grid, dim3(256), 0, st, a);
        else if (axis == 0)                hipLaunchKernelGGL((pair_sum_kernel<LPP, 0, true>), grid, dim3(256), 0, st, a);
        else if (!final_pass)              hipLaunchKernelGGL((pair_sum_kernel<LPP, 1, false>), grid, dim3(256), 0, st, a);
        else                               hipLaunchKernelGGL((pair_sum_kernel<LPP, 1, true>), grid, dim3(256), 0, st, a);
    }
}

// One axis (0 horizontal, 1 vertical).  phase 0: checkpoint pass + sum pass; 1: checkpoint pass only;
// 2: sum pass only (so that the caller can put an event between them).  final_pass: the sum pass adds
// a.Xother and nC*C and writes WTA records instead of X.
void launch_pair(hipStream_t st, const PairArgs& a, int frames, int axis, bool final_pass, int phase) {
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

template <int LPP>
static void launch_sweep_t(hipStream_t st, SweepArgs a, int frames, int mode) {
    constexpr int NWV = FSGM_SWEEP_WAVES;
    constexpr int STRIP = NWV * (64 / LPP), T = (NWV / 2) * (64 / LPP);
    dim3 grid((a.W + STRIP - 1) / STRIP, frames);
    uint8_t* const buf0 = a.state_out;                     // caller passes the base of 2 x frames x state buffers
    uint8_t* const buf1 = a.state_out + (size_t)frames * a.state_frame_stride;
    int b = 0;
    for (int y0 = 0; y0 < a.H; y0 += T, b ^= 1) {
        a.y0 = y0;
        a.rows = T;
        a.state_in = b ? buf0 : buf1;
        a.state_out = b ? buf1 : buf0;
        if (mode == 0)      hipLaunchKernelGGL((sweep_kernel<LPP, 0, NWV>), grid, dim3(NWV * 64), 0, st, a);
        else if (mode == 1) hipLaunchKernelGGL((sweep_kernel<LPP, 1, NWV>), grid, dim3(NWV * 64), 0, st, a);
        else                hipLaunchKernelGGL((sweep_kernel<LPP, 2, NWV>), grid, dim3(NWV * 64), 0, st, a);
    }
}

void launch_sweep(hipStream_t st, const SweepArgs& a, int frames, int mode) {
    switch (agg_packed_lpp(a.D)) {
        case 1: launch_sweep_t<1>(st, a, frames, mode); break;
        case 2: launch_sweep_t<2>(st, a, frames, mode); break;
        case 4: launch_sweep_t<4>(st, a, frames, mode); break;
        case 8: launch_sweep_t<8>(st, a, frames, mode); break;
        case 16: launch_sweep_t<16>(st, a, frames, mode); break;
        default: break;
    }
}

void launch_sweep_finish(hipStream_t st, const WtaArgs& a, const uint4* rec, const uint16_t* s0, int frames) {
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
