"""GPU parity tests for the calc_pyd_cost_sgm path (calc_pyd_cost_sgm.cpp:34-89 sgm_step,
:114-372 sgm2d, :374-437 calc_cost): HIP kernels through the C ABI vs the CPU oracle.
Index / cost outputs bit-exact; mvSub (fp64 parabola) compared exactly as well (same IEEE ops)."""
import numpy as np
import pytest

from fsgm_amd import synth, PydPlan, calc_pyd_cost_sgm
from fsgm_amd._lib import STAGE_COST, STAGE_AGGREGATE, STAGE_WTA

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,mvW,mvH,rX,rY,rAgg,kind", [
    (40, 30, 40, 30, 2, 2, 2, "zero"), (37, 23, 41, 29, 3, 1, 2, "general"), (33, 21, 33, 21, 1, 3, 1, "even"),
    (24, 18, 25, 20, 5, 5, 2, "general"), (16, 12, 16, 12, 0, 0, 0, "general"),
    (45, 19, 45, 19, 5, 5, 2, "int"), (31, 22, 31, 22, 4, 2, 1, "int"), (29, 17, 30, 18, 5, 3, 0, "general"),
    (26, 14, 26, 14, 6, 5, 2, "general"), (26, 14, 26, 14, 2, 6, 2, "int"),       # 13-wide windows: compact layout
    (23, 16, 23, 16, 2, 2, 3, "general"),                                         # aggregation radius 3: generic kernel
    # frames with an interior: waves with every sample inside both images, with samples outside image 2 only, and border waves
    # (the three tap loops of pyd_rows_cost_kernel), at 5, 8 (capped) and 8 pixels a wave
    (160, 90, 160, 90, 5, 5, 2, "general"), (150, 70, 151, 71, 3, 4, 1, "even"), (140, 60, 140, 60, 1, 2, 2, "int"),
])
def test_pyd_cost_volume_bit_exact(gpu_lib, oracle, W, H, mvW, mvH, rX, rY, rAgg, kind):
    I1, I2 = synth.image_pair(W, H, 16, seed=W)
    mv = synth.hint_map(mvW, mvH, kind, seed=H)
    want = oracle.pyd_cost(oracle.census(I1), oracle.census(I2), mv, rAgg, rX, rY)
    with PydPlan(W, H, mvW, mvH, rX, rY, rAgg) as plan:
        plan.upload(0, I1, I2, mv)
        plan.run(STAGE_COST)
        got = plan.download_cost(0)
    np.testing.assert_array_equal(got, want)


AGG = [
    # W, H, rX, rY, P1, P2, cmax, diag, passes, adaptive, hint kind
    (40, 30, 2, 2, 6, 32, 24, 1, 2, 0, "zero"),
    (37, 23, 3, 1, 6, 32, 24, 1, 2, 1, "general"),
    (33, 21, 1, 3, 6, 32, 24, 0, 2, 0, "even"),
    (24, 18, 5, 5, 6, 32, 24, 1, 2, 0, "general"),
    (24, 18, 2, 2, 100, 200, 255, 1, 2, 1, "general"),     # wrapping penalties
    (21, 17, 2, 3, 6, 32, 24, 1, 1, 0, "general"),          # single pass
    (21, 17, 2, 3, 6, 32, 24, 1, 3, 0, "even"),             # third pass repeats the mirrored one
    (9, 1, 2, 2, 6, 32, 24, 1, 2, 0, "general"), (1, 9, 2, 2, 6, 32, 24, 1, 2, 0, "general"),
    # row-packed kernels: every row stride (Sy 1/3, 5/7, 9/11), both shift regimes, shifts beyond the window
    (50, 21, 5, 5, 6, 32, 24, 1, 2, 0, "int"), (50, 21, 5, 5, 6, 32, 24, 1, 2, 1, "even"),
    (35, 18, 5, 4, 6, 64, 24, 1, 2, 0, "int"), (35, 18, 4, 3, 10, 40, 60, 1, 2, 1, "int"),
    (35, 18, 3, 2, 6, 32, 24, 1, 2, 0, "general"), (35, 18, 2, 1, 6, 32, 24, 1, 2, 0, "int"),
    (35, 18, 1, 0, 6, 32, 24, 1, 2, 0, "int"), (35, 18, 0, 5, 6, 32, 24, 0, 2, 0, "int"),
    (19, 33, 5, 5, 0, 0, 24, 1, 2, 0, "int"), (19, 33, 5, 5, 6, 6, 24, 1, 3, 0, "general"),
    (67, 13, 5, 5, 6, 32, 24, 1, 2, 0, "far"),
    # 13-wide windows: compact layout, generic kernels
    (22, 15, 6, 5, 6, 32, 24, 1, 2, 0, "int"), (22, 15, 2, 6, 6, 32, 24, 1, 2, 1, "general"),
]


@pytest.fixture(params=["0", "2"], ids=["packed", "wide-rows"])
def wide_mode(request, monkeypatch):
    """FSGM_PYD_WIDE: both mappings of the row-packed aggregation kernel for the horizontal lines
    (0 = four lines per wave, 2 = one line per wave; the default picks by frame shape and batch)."""
    monkeypatch.setenv("FSGM_PYD_WIDE", request.param)
    return request.param


@pytest.mark.parametrize("W,H,rX,rY,P1,P2,cmax,diag,passes,adaptive,kind", AGG)
def test_pyd_aggregate_and_wta_bit_exact(gpu_lib, oracle, wide_mode, W, H, rX, rY, P1, P2, cmax, diag, passes, adaptive, kind):
    Sx, Sy = 2 * rX + 1, 2 * rY + 1
    I1, I2 = synth.image_pair(W, H, 16, seed=3)
    I1 = (I1.astype(np.int32) * 3 % 256).astype(np.uint8)            # larger gradients: adaptive P2 branch taken
    mv = synth.hint_map(W + 2, H + 1, "int" if kind == "far" else kind, seed=5, amp=14.0 if kind == "far" else 4.0)
    Cv = synth.cost_volume(W, H, Sx * Sy, seed=7, cmax=cmax)
    S = oracle.pyd_aggregate(I1, Cv, mv, Sx, Sy, P1, P2, diag, passes, adaptive)
    bd, mc, ms = oracle.pyd_wta(S, Sx, Sy, 1)
    with PydPlan(W, H, W + 2, H + 1, rX, rY, 2) as plan:
        plan.set_params(P1, P2, diag, passes, adaptive, 1)
        plan.upload(0, I1, I2, mv)
        plan.upload_cost(0, Cv)
        plan.run(STAGE_AGGREGATE | STAGE_WTA)
        gS = plan.download_sum(0)
        gbd, gmc, gms = plan.download(0)
    np.testing.assert_array_equal(gS, S)
    np.testing.assert_array_equal(gbd, bd)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gms, ms)


@pytest.mark.parametrize("W,H,kind,sub", [(64, 48, "zero", 0), (61, 47, "even", 1), (80, 56, "general", 1),
                                          (311, 94, "even", 1), (150, 40, "int", 1)])
def test_calc_pyd_cost_sgm_whole_mex(gpu_lib, oracle, W, H, kind, sub):
    """pyramidal_sgm.m:50 argument values: 5,5 search half sizes, agg 2, P1=6, P2=32, diagonals, 2 passes."""
    I1, I2 = synth.image_pair(W, H, 16, seed=9)
    mv = synth.hint_map(W + 1, H + 1, kind, seed=2)
    bd, mc, ms, Cv, S = oracle.calc_pyd_cost_sgm(I1, I2, mv, 5, 5, 2, sub, 6, 32, 1, 2, 0, want_volumes=True)
    gbd, gmc, gms, gC, gS = calc_pyd_cost_sgm(I1, I2, mv, 5, 5, 2, sub, 6, 32, 1, 2, 0, return_volumes=True)
    np.testing.assert_array_equal(gC, Cv)
    np.testing.assert_array_equal(gS, S)
    np.testing.assert_array_equal(gbd, bd)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gms, ms)
    if not sub:
        assert not gms.any()
