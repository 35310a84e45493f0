// sanitize_driver.cpp -- runs every oracle entry point once under ASan/UBSan (tests/test_oracle_sanitizers.py).
// Inputs include NaN / 1e300 / out-of-image geometry and wrapping penalties.
#include "fsgm_oracle.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main() {
    const int W = 37, H = 23, D = 16;
    std::vector<uint8_t> I1(W*H), I2(W*H);
    for (int i = 0; i < W*H; i++) { I1[i] = rand(); I2[i] = rand(); }
    std::vector<double> pd0(2*W*H), nd(2*W*H), off(W*H);
    for (int i = 0; i < W*H; i++) { pd0[i] = i % W + 1.3; pd0[W*H+i] = i / W + 0.7; nd[i] = 0.6; nd[W*H+i] = -0.8; off[i] = 30 + i % 50; }
    off[5] = 1e300; pd0[7] = 0.0/0.0;
    std::vector<uint32_t> bd(W*H), mc(W*H);
    for (int paths : {4, 8}) fsgm_oracle_calc_cost_sgm(bd.data(), mc.data(), I1.data(), I2.data(), W, H, D, 0.3, pd0.data(), nd.data(), off.data(), 6, 64, paths, nullptr, nullptr);
    fsgm_oracle_calc_cost_sgm(bd.data(), mc.data(), I1.data(), I2.data(), W, H, D, 0.3, pd0.data(), nd.data(), off.data(), 100, 200, 8, nullptr, nullptr);
    const int mvW = W + 2, mvH = H + 1;
    std::vector<double> mv(2*mvW*mvH), mvSub(2*W*H), flow(2*W*H);
    for (auto& v : mv) v = (rand() % 100) / 10.0 - 5.0;
    mv[3] = -1e12; mv[9] = 1e300;
    fsgm_oracle_calc_pyd_cost_sgm(bd.data(), mc.data(), mvSub.data(), I1.data(), I2.data(), W, H, mv.data(), mvW, mvH, 2, 3, 2, 1, 6, 32, 1, 2, 1, nullptr, nullptr);
    fsgm_oracle_calc_pyd_cost_sgm(bd.data(), mc.data(), mvSub.data(), I1.data(), I2.data(), W, H, mv.data(), mvW, mvH, 1, 1, 1, 1, 100, 200, 1, 3, 0, nullptr, nullptr);
    fsgm_oracle_calc_pyd_cost_sgm_ng(mc.data(), flow.data(), I1.data(), I2.data(), W, H, mv.data(), mvW, mvH, 1, 2, 1, 6, 32, nullptr, nullptr);
    std::vector<int32_t> rs(fsgm_oracle_sgm_ng_rand_draws(W, H));
    for (auto& v : rs) v = rand();
    fsgm_oracle_calc_cost_sgm_ng(mc.data(), flow.data(), I1.data(), I2.data(), W, H, 6, 32, rs.data(), rs.size());
    // pyramidal driver, epipolar driver (dense half), post-processing chain
    {
        std::vector<uint8_t> R0(3*W*H), R1(3*W*H);
        for (auto& v : R0) v = rand();
        for (auto& v : R1) v = rand();
        std::vector<double> pmv(2*W*H), f3(3*W*H);
        fsgm_oracle_pyramidal_sgm(pmv.data(), mc.data(), nullptr, R0.data(), R1.data(), W, H, 3, 3, 6, 32, 2, 2, 2, 1, 2, 0);
        fsgm_oracle_pyramidal_sgm(pmv.data(), mc.data(), nullptr, I1.data(), I2.data(), W, H, 1, 6, 6, 32, 1, 1, 1, 1, 2, 1);   // down to 2x1 pixels
        const double F[9] = {0, -1e-3, 0.2, 1e-3, 0, -0.3, -0.2, 0.3, 1}, Hm[9] = {1, 1e-3, 0.5, -1e-3, 1, -0.4, 1e-6, -2e-6, 1};
        std::vector<double> g0(2*W*H), g1(2*W*H), g2(W*H), g3(2*W*H);
        fsgm_oracle_epipolar_maps(g0.data(), g1.data(), g2.data(), g3.data(), F, Hm, 17.3, 11.9, 1, W, H);
        fsgm_oracle_epipolar_sgm_of(f3.data(), mc.data(), R0.data(), R1.data(), W, H, 3, F, Hm, 17.3, 11.9, 0, D, 0.3, 8);
        std::vector<double> D1(W*H), o1(W*H), o2(W*H), o3(W*H);
        for (int i = 0; i < W*H; i++) D1[i] = (rand() % 7 == 0) ? 0.0/0.0 : (rand() % 64) / 4.0;
        fsgm_oracle_postprocess(o1.data(), o2.data(), o3.data(), D1.data(), W, H, pd0.data(), nd.data(), off.data(), 0.3, D + 1, D);
        std::vector<int32_t> lab(W*H);
        fsgm_oracle_speckle_filter(o1.data(), lab.data(), D1.data(), W, H, 0.5, 3);
        fsgm_oracle_vmf(f3.data(), f3.data() + 0, W, H, 0);                      // zero channels: no access
        std::vector<double> med(3*W*H);
        fsgm_oracle_vmf(med.data(), f3.data(), W, H, 3);
    }
    printf("asan/ubsan run finished, checksum %u\n", mc[11]);
    return 0;
}
