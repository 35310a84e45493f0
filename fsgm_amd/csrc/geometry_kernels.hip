// geometry_kernels.hip -- gfx950 kernels for the dense half of the epipolar driver.
// The sparse half of epipolar_geometry.m (SURF features, LMedS fundamental matrix, the two SVDs,
// the expansion vote, :30-96) stays on the host: it produces F, H, the epipole and the direction
// flag, 21 numbers, instead of the 18.6 MB of fp64 maps the MEX is handed per 1242x375 frame.
// Matrix-vector products are written out left to right without FMA contraction (MATLAB hands them
// to BLAS, whose summation order is not specified: the last bit of these maps is not pinned).
#include "geometry_kernels.h"
#include "fsgm_device.h"

namespace fsgm {

__device__ __forceinline__ double row3(const double* m, double x, double y) {      // m0*x + m1*y + m2*1
    return __dadd_rn(__dadd_rn(__dmul_rn(m[0], x), __dmul_rn(m[1], y)), m[2]);
}

__global__ __launch_bounds__(256) void epi_maps_kernel(EpiGeomArgs a) {
    const int xi = blockIdx.x * 64 + (threadIdx.x & 63), yi = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xi >= a.W || yi >= a.H) return;
    const size_t NP = (size_t)a.W * a.H, p = (size_t)yi * a.W + xi;
    const double x = (double)xi, y = (double)yi;                                   // rotation_motion.m:11-13: P0 = (xx-1, yy-1, 1)
    // computeEpipoleLineI2 (rotation_motion.m:48-53)
    double l0 = row3(a.F, x, y), l1 = row3(a.F + 3, x, y), l2 = row3(a.F + 6, x, y);
    double nf = __dsqrt_rn(__dadd_rn(__dmul_rn(l0, l0), __dmul_rn(l1, l1)));
    if (nf < 1e-6) nf = 1.0;
    l0 = __ddiv_rn(l0, nf); l1 = __ddiv_rn(l1, nf); l2 = __ddiv_rn(l2, nf);
    // P1 = H*P0; P1 = P1./P1(3,:)   (:21-22)
    const double q0 = row3(a.Hm, x, y), q1 = row3(a.Hm + 3, x, y), q2 = row3(a.Hm + 6, x, y);
    const double p1x = __ddiv_rn(q0, q2), p1y = __ddiv_rn(q1, q2), p1z = __ddiv_rn(q2, q2);
    double ox = __dsub_rn(p1x, x), oy = __dsub_rn(p1y, y);                         // :23
    const double coef = -__dadd_rn(__dadd_rn(__dmul_rn(l0, p1x), __dmul_rn(l1, p1y)), __dmul_rn(l2, p1z));   // :27
    ox = __dadd_rn(ox, __dmul_rn(coef, l0));                                       // :28
    oy = __dadd_rn(oy, __dmul_rn(coef, l1));
    a.rflow[p] = ox; a.rflow[NP + p] = oy;
    // epipolar_geometry.m:99-115
    const double pdx = __dadd_rn(x + 1.0, ox), pdy = __dadd_rn(y + 1.0, oy);       // :106 PrefD0 = P + Rflow (1-based P)
    a.Pd0[p] = pdx; a.Pd0[NP + p] = pdy;
    double dx = __dsub_rn(pdx, a.ex), dy = __dsub_rn(pdy, a.ey);                   // :107
    if (a.direction) { dx = -dx; dy = -dy; }                                       // :108-110
    const double len = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));   // :112
    a.off[p] = len;
    a.nd[p] = __ddiv_rn(dx, len); a.nd[NP + p] = __ddiv_rn(dy, len);               // :113,:118
}

// epipolar_sgm_of.m:46-51
__global__ __launch_bounds__(256) void epi_flow_kernel(const uint32_t* __restrict__ bestD, const double* __restrict__ nd,
                                                       const double* __restrict__ rflow, double* __restrict__ flow, size_t NP) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const double disp = __ddiv_rn((double)bestD[p], 256.0);                        // :46
    flow[p] = __dadd_rn(__dmul_rn(disp, nd[p]), rflow[p]);                         // :49-50
    flow[NP + p] = __dadd_rn(__dmul_rn(disp, nd[NP + p]), rflow[NP + p]);
    flow[2 * NP + p] = 1.0;                                                        // :51
}

void launch_epi_maps(hipStream_t st, const EpiGeomArgs& a) {
    hipLaunchKernelGGL(epi_maps_kernel, dim3((a.W + 63) / 64, (a.H + 3) / 4), dim3(256), 0, st, a);
}

void launch_epi_flow(hipStream_t st, const uint32_t* bestD, const double* nd, const double* rflow, double* flow, int W, int H) {
    const size_t NP = (size_t)W * H;
    hipLaunchKernelGGL(epi_flow_kernel, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, st, bestD, nd, rflow, flow, NP);
}

}  // namespace fsgm
