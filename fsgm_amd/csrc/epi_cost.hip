// epi_cost.hip -- the cost stage of calc_cost_sgm as ONE kernel: raw census cost along the epipolar line
// (calc_cost_sgm.cpp:343-381) and its 5x5 box mean with replicate border (:387-407) without the raw volume ever
// reaching HBM.  gfx950 only.
//
// Who does what.  A lane owns one pixel column x and 16 consecutive d; wave w of a workgroup owns d = 16 w .. 16 w + 15
// (D / 16 waves: 8 at D = 128), all waves the same 64 columns.  Neighbouring lanes are neighbouring pixels at the same d,
// so the gather of the second image's census word touches two or three cache lines per wave for any smooth direction field
// (the mapping epi_rawcost_px_kernel established), and vzInd(d) sits in scalar registers for the whole kernel.  A workgroup
// walks a strip of 64 raw columns (60 output columns + an apron of 2 on either side) down a segment of rows:
//
//   per row y:   raw costs of the 16 d            the reference's fp64 sequence per voxel, 17 instructions
//                horizontal 5-sum, bytes          raw <= 24, so 5 of them fit a byte: the four neighbours' dwords come through
//                                                 a 1 KB per-wave LDS row and add up 4 costs per instruction (v_add3_u32)
//                split to 2 x u16, into a ring    the last five rows' horizontal sums stay in registers (static slots: the
//                                                 row loop is unrolled five times)
//                vertical 5-sum, 2 x u16          sum of 25 raw costs <= 600
//                mean                             (u8)(1.0 * s / 25 + 0.5) == (2 s + 25) / 50 == round(s / 25) (no ties: 25 is
//                                                 odd) == ONE v_pk_mul_f16 by fp16(0.04) on the sums read as denormal fp16
//                                                 patterns: the product of two fp16 values is exact before its single rounding
//                                                 to the denormal grid (= to an integer, nearest-even), and the relative error
//                                                 of fp16(0.04), 2.1e-4, moves s / 25 <= 24 by at most 0.005 -- the nearest tie
//                                                 is 0.02 away.  Checked for every s < 1024 at plan creation (self-test below)
//                                                 and in tests/test_capi_cpu.py.
//                16 bytes of C per lane           the eight waves' stores fill the pixel's 128-byte line between them
//
// Recomputed apron: 64 / 60 columns x (rows + 4) / rows per segment -- 13 % at 64-row segments, 10 % at 125 -- against a
// second kernel that re-reads the raw volume five rows deep (box5x5_sliding16_kernel: 120 MB per 1242x375x128 frame, 0.035 ms).
#include "epi_kernels.h"
#include "fsgm_device.h"
#include <algorithm>

namespace fsgm {

constexpr int CB_OUT = 60;                    // output columns per workgroup (64 raw columns)
constexpr uint32_t CB_K25 = 0x291F291Fu;      // fp16(0.04) twice

// buffer_load_dword ... idxen: address = base + index * stride (stride 4 from the descriptor), so the gather's address is the
// pixel index itself -- one v_mad_u32_u24 per sample instead of a shift and a multiply-add
typedef int cb_i32x4 __attribute__((ext_vector_type(4)));
__device__ uint32_t cb_struct_buffer_load_u32(cb_i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.i32");
__device__ __forceinline__ cb_i32x4 cb_make_rsrc(const void* base, uint32_t records) {
    const uint64_t b = (uint64_t)base;
    cb_i32x4 r;
    r.x = (int)(uint32_t)b;
    r.y = (int)(((uint32_t)(b >> 32) & 0xFFFFu) | (4u << 16));              // stride 4 bytes in bits 61:48
    r.z = (int)records;                                                      // records of `stride` bytes (idxen range check)
    r.w = 0x00020000;                                                        // gfx9 raw dword format
    return r;
}

__device__ __forceinline__ uint32_t pk_mul_f16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// one row of a lane's per-pixel operands
struct CbPix {
    double px, py, ux, uy, off;
    uint32_t c1;
};

template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void epi_costbox_kernel(EpiCostArgs a, uint8_t* __restrict__ Cout, int seg_rows, uint32_t total_items) {
    __shared__ __attribute__((aligned(16))) uint4 xch[NW][68];              // a wave's raw row: slots 2 .. 65, two spare either side
    __shared__ __attribute__((aligned(16))) uint4 outt[2][64][NW + 1];      // a row of C, [pixel][16-byte piece], one piece of padding
    const int W = a.W, H = a.H, D = a.D;
    // Work item of this workgroup.  Workgroups go to the 8 XCDs round-robin by their linear id; the items (strip fastest, then
    // row segment, then frame) are dealt so that every XCD walks a contiguous eighth of the list: the workgroups resident on an
    // XCD then belong to one or two frames and their samples of image 2's census map share that XCD's L2 (with the plain
    // blockIdx order an XCD sees strips of eight frames at once and a scattered direction field fetches 800 MB per frame).
    const int strips = (W + CB_OUT - 1) / CB_OUT, segs = (H + seg_rows - 1) / seg_rows;
    const uint32_t per_xcd = (total_items + 7u) >> 3;
    const uint32_t item = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || item >= total_items) return;         // uniform over the workgroup
    const int strip = (int)(item % (uint32_t)strips), seg = (int)((item / (uint32_t)strips) % (uint32_t)segs);
    const uint32_t NP = (uint32_t)W * (uint32_t)H;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d0 = 16 * w;
    const size_t f = item / (uint32_t)(strips * segs);
    const int xs = strip * CB_OUT;
    const int xq = xs - 2 + lane;                      // raw column of this lane (may lie outside: replicate)
    const int px = clampi(xq, 0, W - 1);
    const int y0 = seg * seg_rows, y1 = min(y0 + seg_rows, H);
    const int nsteps = y1 - y0 + 4;                                          // raw rows y0 - 2 .. y1 + 1
    const double* __restrict__ p0 = a.pd0 + f * 2 * (size_t)NP;
    const double* __restrict__ nd = a.nd + f * 2 * (size_t)NP;
    const double* __restrict__ of = a.off + f * (size_t)NP;
    const uint32_t* __restrict__ cen1 = a.cen1 + f * (size_t)NP;
    const char* __restrict__ cen2 = (const char*)(a.cen2 + f * (size_t)NP);
    uint8_t* __restrict__ Cf = Cout + f * (size_t)NP * D;
    const int xhi = W - 1, yhi = H - 1;
    const uint32_t W4 = 4u * (uint32_t)W;                                    // (the general path's byte offsets)
    double vz[16];                                                           // wave-uniform: scalar registers
#pragma unroll
    for (int k = 0; k < 16; k++) vz[k] = a.vz[d0 + k];
    const double vzmax = a.vzmax;
    uint4* const myslot = &xch[w][2 + lane];

    auto load_pix = [&](int step) -> CbPix {
        const uint32_t i = (uint32_t)clampi(y0 - 2 + step, 0, yhi) * (uint32_t)W + (uint32_t)px;
        CbPix q;
        q.px = p0[i]; q.py = p0[NP + i]; q.ux = nd[i]; q.uy = nd[NP + i]; q.off = of[i]; q.c1 = cen1[i];
        return q;
    };
    // The fast rounding (round_clamp_small) differs from x86's (int)round(v) only where v + 0.5 reaches 2^31 (cvttsd2si
    // gives INT_MIN there, which clamps to 0; the GPU's convert saturates to INT_MAX, which clamps to hi); NaN and everything
    // negative clamp to 0 on both sides.  A row whose lanes all keep max(bx, by) + |off| max|vz| max(|ux|, |uy|) below 2^30
    // takes it; any other -- including a NaN in that bound -- takes the restated x86 conversion.
    auto row_small = [&](const CbPix& q) -> bool {
        const double bx = __dsub_rn(q.px, 1.0), by = __dsub_rn(q.py, 1.0);
        const double reach = __dmul_rn(__dmul_rn(fabs(q.off), vzmax), fmax(fabs(q.ux), fabs(q.uy)));
        const bool ok = __dadd_rn(fmax(bx, by), reach) < 1073741824.0;
        return __builtin_amdgcn_ballot_w64(!ok) == 0;
    };
    // byte offsets into image 2's census map of the 8 samples d0 + 8 c + k of this lane's pixel (the reference's sequence)
    auto sample_offsets = [&](const int c, const double bx, const double by, const CbPix& q, uint32_t (&boff)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double s = __dmul_rn(q.off, vz[8 * c + k]);                                  // offset * vzInd      :360-364
            const double ox = __dmul_rn(s, q.ux), oy = __dmul_rn(s, q.uy);                      // :365-366
            const double vx = __dadd_rn(bx, ox), vy = __dadd_rn(by, oy);
            const int x2 = round_clamp_small(vx, xhi), y2 = round_clamp_small(vy, yhi);         // :371-375
            boff[k] = __umul24((uint32_t)y2, (uint32_t)W) + (uint32_t)x2;                       // pixel index (W < 2^22, H < 2^24: launcher)
        }
    };
    const cb_i32x4 cen2_rsrc = cb_make_rsrc(cen2, NP);
    auto gather = [&](const uint32_t (&idx)[8], uint32_t (&word)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) word[k] = cb_struct_buffer_load_u32(cen2_rsrc, (int)idx[k], 0, 0, 0);
    };
    auto hamming = [&](const uint32_t c1, const uint32_t (&word)[8], uint32_t& lo, uint32_t& hi) {
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = __popc(c1 ^ word[k]);                                // :377-378
        auto pack4 = [](uint32_t c0, uint32_t c1_, uint32_t c2, uint32_t c3) {
            uint32_t t, u, r;
            asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(t) : "v"(c1_), "v"(c0));
            asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(u) : "v"(c3), "v"(c2));
            asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r) : "v"(u), "v"(t));
            return r;
        };
        lo = pack4(c[0], c[1], c[2], c[3]); hi = pack4(c[4], c[5], c[6], c[7]);
    };

    uint32_t ring[5][8];                                                     // horizontal 5-sums of the last five rows, 2 x u16
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
        for (int i = 0; i < 8; i++) ring[s][i] = 0;

    // A step = the fill of raw row `st` (st < nsteps) with the tail of row st - 1 tucked between the gathers' issue and their
    // use: horizontal sums through LDS, ring update, vertical sum, mean, store of output row y0 + st - 5.  Software pipeline:
    // the next row's operands are requested once this row's offsets are out and used a step later; no register double-buffering.
    CbPix cur = load_pix(0);
    uint4 raw = make_uint4(0, 0, 0, 0);

    auto tail = [&](const int st, uint32_t (&slot)[8]) {                     // raw = costs of raw row st
        // horizontal 5-sum across the lanes, in bytes (<= 120)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        *myslot = raw;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint4 m2 = myslot[-2], m1 = myslot[-1], p1 = myslot[1], p2 = myslot[2];
        __builtin_amdgcn_wave_barrier();
        const uint32_t h[4] = {raw.x + m2.x + m1.x + p1.x + p2.x, raw.y + m2.y + m1.y + p1.y + p2.y,
                               raw.z + m2.z + m1.z + p1.z + p2.z, raw.w + m2.w + m1.w + p1.w + p2.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            slot[2 * i] = h[i] & 0x00FF00FFu;                                // d = 4 i, 4 i + 2
            slot[2 * i + 1] = __builtin_amdgcn_perm(0u, h[i], 0x0C030C01u);  // d = 4 i + 1, 4 i + 3
        }
        const int yo = y0 + st - 4;                                          // the row whose five raw rows are now in the ring
        if (st >= 4) {                                                       // wave-uniform
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t se = ring[0][2 * i] + ring[1][2 * i] + ring[2][2 * i] + ring[3][2 * i] + ring[4][2 * i];
                const uint32_t so = ring[0][2 * i + 1] + ring[1][2 * i + 1] + ring[2][2 * i + 1] + ring[3][2 * i + 1] + ring[4][2 * i + 1];
                const uint32_t me = pk_mul_f16(se, CB_K25), mo = pk_mul_f16(so, CB_K25);   // :403-404
                asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(o[i]) : "v"(mo), "v"(me));
            }
            // the waves' 16-byte pieces of a pixel meet in LDS and leave as whole lines: wave w stores pixels 64 / NW * w ...,
            // one 16-byte piece per lane (16-byte stores 128 bytes apart were measured at 1.85 x the written bytes in HBM)
            outt[st & 1][lane][w] = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (st >= 4) {
            __syncthreads();                                                 // every wave runs the same steps: uniform
            constexpr int PPW = 64 / NW;                                     // pixels a wave stores
            const int pl = w * PPW + lane / NW, piece = lane % NW;           // pixel (= lane of the producers), piece
            const int xo = xs - 2 + pl;
            if (pl >= 2 && pl < 2 + CB_OUT && xo < W)
                *(uint4*)(Cf + ((uint32_t)yo * (uint32_t)W + (uint32_t)xo) * (uint32_t)D + 16u * piece) = outt[st & 1][pl][piece];
        }
    };
    auto fill_general = [&](const int st, uint32_t (&pslot)[8]) {
        if (st >= 1) tail(st - 1, pslot);                                    // wave-uniform
        const double bx = __dsub_rn(cur.px, 1.0), by = __dsub_rn(cur.py, 1.0);
        const double* vzp = a.vz + d0;
#pragma clang loop unroll(disable)
        for (int k = 0; k < 16; k++) {
            const double s = __dmul_rn(cur.off, vzp[k]);
            const double ox = __dmul_rn(s, cur.ux), oy = __dmul_rn(s, cur.uy);
            const int x2 = clamp0(round_to_i32_x86(__dadd_rn(bx, ox)), xhi), y2 = clamp0(round_to_i32_x86(__dadd_rn(by, oy)), yhi);
            const uint32_t cost = __popc(cur.c1 ^ *(const uint32_t*)(cen2 + (__umul24((uint32_t)y2, W4) + ((uint32_t)x2 << 2))));
            raw.x = __builtin_amdgcn_alignbit(raw.y, raw.x, 8);              // the 16 bytes as one shift register: byte k enters at the top
            raw.y = __builtin_amdgcn_alignbit(raw.z, raw.y, 8);
            raw.z = __builtin_amdgcn_alignbit(raw.w, raw.z, 8);
            raw.w = (raw.w >> 8) | (cost << 24);
        }
        cur = load_pix(min(st + 1, nsteps - 1));
    };
    // xonly: every lane of the row has a direction with uy == 0 (horizontal epipolar lines: a rectified pair, the survey's
    // timing maps) -- offset * vzInd * 0 is a zero of either sign for the finite products a small row has, by + (+-0) is by,
    // so the sample row is round(by) for every d (:372, :375) and only the x coordinate walks: 11 instructions a voxel, not 17.
    // (A wave-uniform branch around the two forms of the offsets only: a third copy of the whole step cost the register
    // allocation 54 spills.)
    auto sample_offsets_x = [&](const int c, const double bx, const uint32_t rowoff, const CbPix& q, uint32_t (&boff)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double s = __dmul_rn(q.off, vz[8 * c + k]);
            const double vx = __dadd_rn(bx, __dmul_rn(s, q.ux));
            boff[k] = rowoff + (uint32_t)round_clamp_small(vx, xhi);
        }
    };
    auto fill = [&](const int st, uint32_t (&pslot)[8]) {
        const double bx = __dsub_rn(cur.px, 1.0), by = __dsub_rn(cur.py, 1.0);                 // :348-349
        const bool xonly = __builtin_amdgcn_ballot_w64(cur.uy != 0.0) == 0;                    // (+-0 compare equal to 0; NaN does not)
        const uint32_t rowoff = __umul24((uint32_t)round_clamp_small(by, yhi), (uint32_t)W);
        uint32_t w0[8], w1[8];
        if (xonly) sample_offsets_x(0, bx, rowoff, cur, w0); else sample_offsets(0, bx, by, cur, w0);
        gather(w0, w0);
        if (xonly) sample_offsets_x(1, bx, rowoff, cur, w1); else sample_offsets(1, bx, by, cur, w1);
        gather(w1, w1);
        const uint32_t c1 = cur.c1;
        cur = load_pix(min(st + 1, nsteps - 1));                             // (past the last row: the last row again, never used)
        if (st >= 1) tail(st - 1, pslot);                                    // wave-uniform
        hamming(c1, w0, raw.x, raw.y);
        hamming(c1, w1, raw.z, raw.w);
    };

    for (int base = 0; base < nsteps; base += 5) {
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (base + k < nsteps) {                                         // wave-uniform
                if (row_small(cur)) fill(base + k, ring[(k + 4) % 5]);
                else                fill_general(base + k, ring[(k + 4) % 5]);
            }
    }
    // the last raw row's tail
    switch ((nsteps - 1) % 5) {
        case 0: tail(nsteps - 1, ring[0]); break;
        case 1: tail(nsteps - 1, ring[1]); break;
        case 2: tail(nsteps - 1, ring[2]); break;
        case 3: tail(nsteps - 1, ring[3]); break;
        default: tail(nsteps - 1, ring[4]); break;
    }
}

// self-test of the mean's multiply (every sum a 5x5 window of census costs can reach, and beyond): 0 = exact
__global__ void costbox_selftest_kernel(uint32_t* bad) {
    const uint32_t s = threadIdx.x + 256u * blockIdx.x;                      // 0 .. 1023
    const uint32_t got = pk_mul_f16(s | (s << 16), CB_K25);
    const uint32_t want = (2 * s + 25) / 50;
    if ((got & 0xFFFFu) != want || (got >> 16) != want) atomicAdd(bad, 1u);
}

int costbox_selftest(hipStream_t st) {
    uint32_t* d = nullptr;
    uint32_t h = 1;
    if (hipMalloc((void**)&d, 4) != hipSuccess) return -1;
    bool ok = hipMemsetAsync(d, 0, 4, st) == hipSuccess;
    if (ok) hipLaunchKernelGGL(costbox_selftest_kernel, dim3(4), dim3(256), 0, st, d);
    ok = ok && hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    (void)hipFree(d);
    return ok ? (int)h : -1;
}

bool costbox_ok(int W, int H, int D) {
    return (D == 16 || D == 32 || D == 64 || D == 128 || D == 256) && W < (1 << 22) && H < (1 << 24);
}

// Rows per segment.  A workgroup's time goes with its rows + 4 (the apron rows), the launch's with the rounds of resident
// workgroups it takes (16 waves per CU at <= 128 VGPRs: 16 / NW workgroups): the segment count that minimises
// rounds x (rows + 4) -- few long segments once the frames alone fill the chip, many short ones for a single frame.
int costbox_seg_rows(int W, int H, int D, int frames, int cus) {
    const long long strips = (W + CB_OUT - 1) / CB_OUT;
    const long long resident = (long long)cus * std::max(1, 16 / (D >> 4));
    long long best = -1;
    int best_rows = H;
    for (int nseg = 1; nseg <= std::max(1, H / 8); nseg++) {
        const int rows = (H + nseg - 1) / nseg;                              // equal segments: 375 rows -> 3 x 125, not 128 + 128 + 119
        const long long items = strips * ((H + rows - 1) / rows) * frames;
        const long long cost = ((items + resident - 1) / resident) * (rows + 4);
        if (best < 0 || cost < best) { best = cost; best_rows = rows; }
    }
    return best_rows;
}

void launch_epi_costbox(hipStream_t st, const EpiCostArgs& a, uint8_t* C, int frames) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const int seg = costbox_seg_rows(a.W, a.H, a.D, frames, cus);
    const uint32_t total = (uint32_t)((a.W + CB_OUT - 1) / CB_OUT) * (uint32_t)((a.H + seg - 1) / seg) * (uint32_t)frames;
    dim3 grid(8u * ((total + 7u) / 8u));
    switch (a.D >> 4) {
#define FSGM_CB(NW) case NW: hipLaunchKernelGGL(epi_costbox_kernel<NW>, grid, dim3(NW * 64), 0, st, a, C, seg, total); break;
        FSGM_CB(1) FSGM_CB(2) FSGM_CB(4) FSGM_CB(8) FSGM_CB(16)
#undef FSGM_CB
    }
}

}  // namespace fsgm
