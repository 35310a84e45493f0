// pyramid_kernels.h -- launch interface of the pyramidal_sgm.m level-loop kernels
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace fsgm {

struct PyrFlowArgs {
    const uint32_t* bestD;  // [H][W] candidate index sx*Sy + sy
    const double* mvSub;    // [2][H][W]
    const double* mvPre;    // [2][mvH][mvW] the hint map this level ran with
    double* flow;           // [2][H][W] this level's flow (mvPyd{l})
    double* next;           // [2][2H][2W] hint map of the next finer level, or null at level 1
    int W, H, mvW, mvH;
    int Sy, hor, ver;
    size_t next_frame_stride;   // doubles between frames of `next` (2 * its map size); frames of the other arrays are contiguous
};

// impyramid 'reduce' of `planes` images [planes][H][W] -> [planes][ceil(H/2)][ceil(W/2)]
void launch_pyr_reduce(hipStream_t st, const uint8_t* in, uint8_t* out, int W, int H, int planes);
void launch_pyr_gray(hipStream_t st, const uint8_t* rgb, uint8_t* out, int W, int H, int frames = 1);   // rgb [frames][3][H][W] -> [frames][H][W]
void launch_pyr_flow(hipStream_t st, const PyrFlowArgs& a, int frames = 1);   // every array [frames][...]
// next[2][2H][2W] = 2 * flow[2][H][W] at (y/2, x/2): 2*imresize(mv, 2, 'nearest') (pyramidal_sgm.m:72)
// (flow [frames][2][H][W]; frame f of `next` starts next_frame_stride doubles after frame f-1)
void launch_pyr_upsample2(hipStream_t st, const double* flow, double* next, int W, int H, int frames = 1, size_t next_frame_stride = 0);

}  // namespace fsgm
