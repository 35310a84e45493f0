// post_kernels.h -- launch interface of the post-processing kernels (the reference's MATLAB functions
// speckle_filter.m, calc_disp_from_first.m, forward_backward_check.m, scanline_in_fill.m, vzInd2Disp.m;
// chained by test.m:45-50).  Maps are f64 [H][W], NaN = invalid, x fastest.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace fsgm {

struct PostGeom {           // the three maps of epipolar_geometry.m that the disparity functions use
    const double* Pd0;      // [2][H][W], 1-based pixel coordinates, plane 0 = x
    const double* nd;       // [2][H][W]
    const double* O;        // [H][W]
    double vMax, n;
};

// speckle_filter.m: out = image with every 4-connected region (neighbours joined when both valid and
// |a-b| < maxDiff) of fewer than maxSpeckleSize pixels set to NaN.  parent, size: i32 [H*W] scratch
// (parent ends up holding each pixel's region root = the region's first pixel in raster order).
// labels (may be null): i32 [H*W], regions numbered in raster order of their first pixel, 0 = invalid;
// scan: i32 [H*W/1024 + 2] scratch for it.
void launch_speckle_filter(hipStream_t st, const double* image, double* out, int32_t* labels, int32_t* parent,
                           int32_t* size, int32_t* scan, int W, int H, double maxDiff, double maxSpeckleSize);
void launch_disp_from_first(hipStream_t st, const double* D1, double* D2, const PostGeom& g, int W, int H);
void launch_fb_check(hipStream_t st, const double* D1, const double* D2, double* out, const PostGeom& g, int W, int H);
// left: i32 [H*W] scratch
void launch_scanline_in_fill(hipStream_t st, const double* in, double* out, int32_t* left, int W, int H);
void launch_vmf(hipStream_t st, const double* in, double* out, int W, int H, int channels);   // vmf.m: 5x5 median per plane
void launch_vzind2disp(hipStream_t st, const double* w, const double* O, double* D, size_t n_px, double vMax, double n);

}  // namespace fsgm
