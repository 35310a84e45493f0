// geometry_kernels.h -- the dense half of the epipolar driver: per-pixel maps from the sparse geometry
// (epipolar_geometry.m:99-115, rotation_motion.m) and the flow composition of epipolar_sgm_of.m:46-51
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace fsgm {

struct EpiGeomArgs {
    double F[9], Hm[9];     // row-major 3x3: fundamental matrix, rotation homography K*R/K
    double ex, ey;          // epipole in image 2
    int direction;          // 1 = contraction: directions point towards the epipole (epipolar_geometry.m:109-111)
    double* Pd0;            // [2][H][W] out, 1-based coordinates
    double* nd;             // [2][H][W] out
    double* off;            // [H][W] out
    double* rflow;          // [2][H][W] out
    int W, H;
};

void launch_epi_maps(hipStream_t st, const EpiGeomArgs& a);
// flow [3][H][W]: (bestD/256) * nd + rflow, third plane 1
void launch_epi_flow(hipStream_t st, const uint32_t* bestD, const double* nd, const double* rflow, double* flow, int W, int H);

}  // namespace fsgm
