"""Seeded random-configuration parity sweep: GPU (through the C ABI) vs CPU oracle over random
shapes, disparity ranges, penalties, hint maps and switches -- shapes that are not multiples of
any tile (strip, wave, row block) and parameters on both sides of the no-wrap boundary."""
import os

import numpy as np
import pytest

from fsgm_amd import synth, EpiPlan, PydPlan, calc_pyd_cost_sgm_ng, calc_cost_sgm_ng
from fsgm_amd._lib import STAGE_AGGREGATE, STAGE_WTA, STAGE_ALL

pytestmark = pytest.mark.gpu

# FSGM_FUZZ_SEEDS=N in the environment runs every sweep below with N seeds instead of its default count (a soak run)
_SOAK = int(os.environ.get("FSGM_FUZZ_SEEDS", "0"))


def _seeds(default):
    return range(_SOAK if _SOAK > 0 else default)


def _rng(seed):
    return np.random.RandomState(seed)          # only picks test configurations; data comes from synth


@pytest.mark.parametrize("seed", _seeds(24))
def test_epi_random_configs(gpu_lib, oracle, seed):
    r = _rng(seed)
    D = int(r.choice([16, 32, 64, 128, 256, 8, 20, 48, 100]))
    W, H = int(r.randint(1, 90)), int(r.randint(1, 70))
    paths = int(r.choice([4, 8]))
    if r.rand() < 0.6:
        P1, P2, cmax = int(r.randint(0, 20)), int(r.randint(0, 86)), 24        # no-wrap side
    else:
        P1, P2, cmax = int(r.randint(0, 256)), int(r.randint(0, 256)), int(r.choice([24, 255]))
    sub, vz = int(r.rand() < 0.7), int(r.rand() < 0.5)
    B = int(r.choice([1, 2, 5]))
    vols = [synth.cost_volume(W, H, D, seed=seed * 10 + f, cmax=cmax) for f in range(B)]
    _, _, off = synth.epi_maps(W, H, "general", seed=seed)
    with EpiPlan(W, H, D, B, paths=paths, subpixel=sub, vz_to_disp=vz) as plan:
        plan.set_penalties(P1, P2, 0.3)
        for f in range(B):
            plan.upload_cost(f, vols[f])
            plan.upload_offset(f, off)
        for mode in (1, 2, 3, 6, 4, 5):                      # a mode whose pipeline does not cover the configuration falls back
            plan.set_agg_mode(mode)
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
            for f in range(B):
                S = oracle.epi_aggregate(vols[f], P1, P2, paths)
                bd, mc = oracle.epi_wta(S, W, H, D, sub)
                if vz:
                    bd = oracle.epi_vz_to_disp(bd, off, 0.3, D + 1)
                gbd, gmc = plan.download(f)
                msg = f"seed {seed} mode {mode} {plan.kernel_name} W{W} H{H} D{D} paths{paths} P{P1},{P2} frame {f}"
                np.testing.assert_array_equal(gmc, mc, err_msg=msg)
                np.testing.assert_array_equal(gbd, bd, err_msg=msg)
                np.testing.assert_array_equal(plan.download_sum(f), S[:-1].reshape(H, W, D), err_msg=msg)


@pytest.mark.parametrize("seed", _seeds(16))
def test_epi_random_tall_configs(gpu_lib, oracle, seed):
    """Tall narrow frames: several bands of the band sweeps (both forms), several row blocks of the block sweeps."""
    r = _rng(500 + seed)
    D = int(r.choice([16, 32, 64, 128, 128, 256]))
    W, H = int(r.randint(1, 40)), int(r.randint(60, 330))
    paths = int(r.choice([4, 8]))
    P1 = int(r.randint(0, 30))
    P2 = int(r.randint(P1, 100))                            # P1 <= P2; some beyond the fused kernels' byte budgets
    sub, vz = int(r.rand() < 0.7), int(r.rand() < 0.5)
    B = int(r.choice([1, 2, 3]))
    vols = [synth.cost_volume(W, H, D, seed=seed * 10 + f, cmax=24) for f in range(B)]
    for v in vols:
        v[:, ::4, :] = 0
    _, _, off = synth.epi_maps(W, H, "general", seed=seed)
    want = []
    for f in range(B):
        S = oracle.epi_aggregate(vols[f], P1, P2, paths)
        bd, mc = oracle.epi_wta(S, W, H, D, sub)
        want.append((oracle.epi_vz_to_disp(bd, off, 0.3, D + 1) if vz else bd, mc, S[:-1].reshape(H, W, D)))
    with EpiPlan(W, H, D, B, paths=paths, subpixel=sub, vz_to_disp=vz) as plan:
        plan.set_penalties(P1, P2, 0.3)
        for f in range(B):
            plan.upload_cost(f, vols[f])
            plan.upload_offset(f, off)
        for mode in (4, 5, 2, 6, 1, 5, 4):
            plan.set_agg_mode(mode)
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
            for f in range(B):
                gbd, gmc = plan.download(f)
                msg = f"seed {seed} mode {mode} {plan.kernel_name} W{W} H{H} D{D} paths{paths} P{P1},{P2} frame {f}"
                np.testing.assert_array_equal(gmc, want[f][1], err_msg=msg)
                np.testing.assert_array_equal(gbd, want[f][0], err_msg=msg)
            np.testing.assert_array_equal(plan.download_sum(B - 1), want[B - 1][2], err_msg=f"seed {seed} mode {mode} S")
        plan.sync()


@pytest.mark.parametrize("seed", _seeds(12))
def test_pyd_random_configs(gpu_lib, oracle, seed):
    r = _rng(100 + seed)
    W, H = int(r.randint(1, 60)), int(r.randint(1, 45))
    rX, rY, rAgg = int(r.randint(0, 6)), int(r.randint(0, 6)), int(r.randint(0, 4))
    mvW, mvH = W + int(r.randint(0, 4)), H + int(r.randint(0, 4))
    kind = str(r.choice(["zero", "even", "general"]))
    P1, P2 = (6, 32) if r.rand() < 0.6 else (int(r.randint(0, 256)), int(r.randint(0, 256)))
    diag, passes, adaptive, sub = int(r.rand() < 0.7), int(r.choice([1, 2, 2, 3])), int(r.rand() < 0.5), int(r.rand() < 0.5)
    I1, I2 = synth.image_pair(W, H, 16, seed=seed)
    I1 = (I1.astype(np.int32) * 3 % 256).astype(np.uint8)
    mv = synth.hint_map(mvW, mvH, kind, seed=seed, amp=float(r.choice([1.5, 4.0, 9.0])))
    bd, mc, ms, Cv, S = oracle.calc_pyd_cost_sgm(I1, I2, mv, rX, rY, rAgg, sub, P1, P2, diag, passes, adaptive, want_volumes=True)
    with PydPlan(W, H, mvW, mvH, rX, rY, rAgg) as plan:
        plan.set_params(P1, P2, diag, passes, adaptive, sub)
        plan.upload(0, I1, I2, mv)
        plan.run(STAGE_ALL)
        msg = f"seed {seed} W{W} H{H} r{rX},{rY},{rAgg} {kind} P{P1},{P2} diag{diag} passes{passes} ad{adaptive}"
        np.testing.assert_array_equal(plan.download_cost(0), Cv, err_msg=msg)
        np.testing.assert_array_equal(plan.download_sum(0), S, err_msg=msg)
        gbd, gmc, gms = plan.download(0)
    np.testing.assert_array_equal(gbd, bd, err_msg=msg)
    np.testing.assert_array_equal(gmc, mc, err_msg=msg)
    np.testing.assert_array_equal(gms, ms, err_msg=msg)


@pytest.mark.parametrize("seed", _seeds(16))
def test_ng_random_configs(gpu_lib, oracle, seed):
    r = _rng(200 + seed)
    W, H = int(r.randint(1, 40)), int(r.randint(1, 30))
    mvW, mvH = int(r.randint(1, W + 3)), int(r.randint(1, H + 3))
    P1, P2 = (6, 32) if r.rand() < 0.5 else (int(r.randint(0, 256)), int(r.randint(0, 256)))
    half, agg, sub = int(r.choice([0, 1, 1, 2])), int(r.randint(0, 6)), int(r.rand() < 0.5)
    I1, I2 = synth.image_pair(W, H, 16, seed=seed + 50)
    kind, amp = str(r.choice(["zero", "even", "int", "general"])), float(r.choice([0.7, 2.0, 6.0]))   # few to many repeated candidates
    mv = synth.hint_map(mvW, mvH, kind, seed=seed, amp=amp)
    mc, fl, _, S = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, half, agg, sub, P1, P2, want_volumes=True)
    gmc, gfl, gS = calc_pyd_cost_sgm_ng(I1, I2, mv, half, agg, sub, P1, P2, return_sum=True)
    msg = f"seed {seed} W{W} H{H} mv{mvW}x{mvH} half{half} agg{agg} P{P1},{P2} {kind} amp{amp}"
    np.testing.assert_array_equal(gS, S, err_msg=msg)
    np.testing.assert_array_equal(gmc, mc, err_msg=msg)
    np.testing.assert_array_equal(gfl, fl, err_msg=msg)
    W2, H2 = min(W, 24), min(H, 16)
    I1, I2 = np.ascontiguousarray(I1[:H2, :W2]), np.ascontiguousarray(I2[:H2, :W2])
    rs = oracle.glibc_rand_stream(oracle.sgm_ng_rand_draws(W2, H2), seed=seed + 1)
    mc, fl = oracle.calc_cost_sgm_ng(I1, I2, P1, P2, rs)
    gmc, gfl = calc_cost_sgm_ng(I1, I2, None, 1, 2, 0, P1, P2, rand_stream=rs)
    np.testing.assert_array_equal(gmc, mc, err_msg=msg)
    np.testing.assert_array_equal(gfl, fl, err_msg=msg)
