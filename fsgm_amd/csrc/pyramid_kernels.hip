// pyramid_kernels.hip -- gfx950 kernels for the level loop of pyramidal_sgm.m around
// calc_pyd_cost_sgm (reference: pyramidal_sgm.m, cited per kernel).  impyramid, rgb2gray and
// imresize are MATLAB toolbox functions outside the reference tree; they follow the published
// behaviour written down in oracle/fsgm_oracle_pyramid.cpp (parity unpinned there).
#include "pyramid_kernels.h"
#include "fsgm_device.h"

namespace fsgm {

// imresize's border rule: index table [1:n n:-1:1], i.e. mirror with the edge sample repeated
__device__ __forceinline__ int mirror_idx(int i, int n) {
    const int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}
__device__ __forceinline__ int tap5(int a, int b, int c, int d, int e) { return (a + 4 * b + 6 * c + 4 * d + e + 8) >> 4; }

// impyramid(A, 'reduce') (pyramidal_sgm.m:28-31): weights [1 4 6 4 1]/16 on input samples 2i-2..2i+2,
// rows first, the intermediate rounded to uint8, then columns.  One thread per output sample; the
// five intermediate samples it needs are recomputed (25 loads from a tiny image).
__global__ __launch_bounds__(256) void pyr_reduce_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                         int W, int H, int W2, int H2) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= W2 || i >= H2) return;
    const uint8_t* src = in + (size_t)blockIdx.z * W * H;
    int rows[5];
#pragma unroll
    for (int k = 0; k < 5; k++) rows[k] = mirror_idx(2 * i - 2 + k, H) * W;
    int t[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int x = mirror_idx(2 * j - 2 + k, W);
        t[k] = tap5(src[rows[0] + x], src[rows[1] + x], src[rows[2] + x], src[rows[3] + x], src[rows[4] + x]);
    }
    out[(size_t)blockIdx.z * W2 * H2 + (size_t)i * W2 + j] = (uint8_t)tap5(t[0], t[1], t[2], t[3], t[4]);
}

// rgb2gray (pyramidal_sgm.m:44-45): round(0.2989 R + 0.5870 G + 0.1140 B), the toolbox's coefficients
__global__ __launch_bounds__(256) void pyr_gray_kernel(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    rgb += (size_t)blockIdx.y * 3 * n;                       // frame
    out += (size_t)blockIdx.y * n;
    const double v = __dadd_rn(__dadd_rn(__dmul_rn(0.298936021293775, (double)rgb[i]), __dmul_rn(0.587043074451121, (double)rgb[n + i])),
                               __dmul_rn(0.114020904255103, (double)rgb[2 * (size_t)n + i]));
    out[i] = (uint8_t)(int)floor(__dadd_rn(v, 0.5));
}

// pyramidal_sgm.m:57-72: index -> (mvx, mvy) (ind2sub over [2*ver+1, 2*hor+1]), plus the hint the level
// started from (its own stride) plus the sub-pixel part; then the next finer level's hint map
// 2*imresize(mv, 2, 'nearest') = every value doubled and written to a 2x2 block.
__global__ __launch_bounds__(256) void pyr_flow_kernel(PyrFlowArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const size_t n = (size_t)a.W * a.H, i = (size_t)y * a.W + x, ip = (size_t)y * a.mvW + x, nm = (size_t)a.mvW * a.mvH;
    const size_t f = blockIdx.z;                             // frame
    a.bestD += f * n; a.mvSub += f * 2 * n; a.mvPre += f * 2 * nm; a.flow += f * 2 * n;
    if (a.next) a.next += f * a.next_frame_stride;
    const uint32_t best = a.bestD[i];
    const int sx = (int)(best / (uint32_t)a.Sy), sy = (int)(best % (uint32_t)a.Sy);
    const double vx = __dadd_rn(__dadd_rn((double)(sx - a.hor), a.mvPre[ip]), a.mvSub[i]);               // :59,:64
    const double vy = __dadd_rn(__dadd_rn((double)(sy - a.ver), a.mvPre[nm + ip]), a.mvSub[n + i]);      // :60,:64
    a.flow[i] = vx;
    a.flow[n + i] = vy;
    if (a.next) {                                                                                        // :72
        const size_t W2 = 2 * (size_t)a.W, n2 = 4 * n, o = (size_t)(2 * y) * W2 + 2 * x;
        const double2 dx = make_double2(2.0 * vx, 2.0 * vx), dy = make_double2(2.0 * vy, 2.0 * vy);
        *(double2*)(a.next + o) = dx;
        *(double2*)(a.next + o + W2) = dx;
        *(double2*)(a.next + n2 + o) = dy;
        *(double2*)(a.next + n2 + o + W2) = dy;
    }
}

void launch_pyr_reduce(hipStream_t st, const uint8_t* in, uint8_t* out, int W, int H, int planes) {
    const int W2 = (W + 1) / 2, H2 = (H + 1) / 2;
    dim3 grid((W2 + 63) / 64, (H2 + 3) / 4, planes);
    hipLaunchKernelGGL(pyr_reduce_kernel, grid, dim3(256), 0, st, in, out, W, H, W2, H2);
}

void launch_pyr_gray(hipStream_t st, const uint8_t* rgb, uint8_t* out, int W, int H, int frames) {
    const int n = W * H;
    hipLaunchKernelGGL(pyr_gray_kernel, dim3((n + 255) / 256, frames), dim3(256), 0, st, rgb, out, n);
}

void launch_pyr_flow(hipStream_t st, const PyrFlowArgs& a, int frames) {
    dim3 grid((a.W + 63) / 64, (a.H + 3) / 4, frames);
    hipLaunchKernelGGL(pyr_flow_kernel, grid, dim3(256), 0, st, a);
}

__global__ __launch_bounds__(256) void pyr_upsample2_kernel(const double* __restrict__ flow, double* __restrict__ next, int W, int H,
                                                            size_t next_frame_stride) {
    const int W2 = 2 * W, H2 = 2 * H;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W2 * H2) return;
    const int y = i / W2, x = i - y * W2;
    const size_t src = (size_t)(y >> 1) * W + (x >> 1), np = (size_t)W * H, np2 = (size_t)W2 * H2;
    flow += blockIdx.y * 2 * np;
    next += blockIdx.y * next_frame_stride;
    next[i] = __dmul_rn(2.0, flow[src]);
    next[np2 + i] = __dmul_rn(2.0, flow[np + src]);
}

void launch_pyr_upsample2(hipStream_t st, const double* flow, double* next, int W, int H, int frames, size_t next_frame_stride) {
    hipLaunchKernelGGL(pyr_upsample2_kernel, dim3((4 * W * H + 255) / 256, frames), dim3(256), 0, st, flow, next, W, H, next_frame_stride);
}

}  // namespace fsgm
