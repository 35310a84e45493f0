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
    printf("asan/ubsan run finished, checksum %u\n", mc[11]);
    return 0;
}
