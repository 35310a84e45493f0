// epi_wta_tail.h -- sub-pixel + vz->disparity tail shared by the WTA kernels
// (calc_cost_sgm.cpp:278-308, :414-426).  c_1, c, c1 are S[best-1], S[best], S[best+1].
#pragma once
#include "epi_kernels.h"
#include "fsgm_device.h"

namespace fsgm {

__device__ __forceinline__ void wta_finish(const WtaArgs& a, size_t f, int p, uint32_t best, uint32_t minc,
                                           uint32_t c_1, uint32_t c1) {
    const size_t NP = (size_t)a.W * a.H;
    uint32_t bd = best;
    if (a.subpixel) {
        if (best > 1 && best < (uint32_t)a.D) {                                   // :293
            const double dc_1 = (double)c_1, dc = (double)minc, dc1 = (double)c1;
            double sub = (double)best;
            if (dc1 < dc_1) sub = __dadd_rn(sub, __ddiv_rn(__ddiv_rn(__dsub_rn(dc1, dc_1), __dsub_rn(dc, dc_1)), 2.0));   // :299
            else            sub = __dadd_rn(sub, __ddiv_rn(__ddiv_rn(__dsub_rn(dc1, dc_1), __dsub_rn(dc, dc1)), 2.0));    // :301
            bd = f64_to_u32_x86(__dmul_rn(sub, 256.0));                           // :303
        } else {
            bd = best * 256u;                                                     // :305
        }
    }
    if (a.vz_to_disp) {                                                           // :414-426
        const double d = __ddiv_rn((double)bd, 256.0);
        const double r = __dmul_rn(__ddiv_rn(d, (double)(a.D + 1)), a.vMax);
        const double vz = __ddiv_rn(r, __dsub_rn(1.0, r));
        bd = f64_to_u32_x86(__dmul_rn(__dmul_rn(a.off[f * NP + p], vz), 256.0));
    }
    a.bestD[f * NP + p] = bd;
    a.minC[f * NP + p] = minc;
}


}  // namespace fsgm
