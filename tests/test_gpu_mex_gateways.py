"""The mexFunction gateways called the way the MATLAB drivers call them, vs the CPU oracle.
epipolar_sgm_of.m:45, pyramidal_sgm.m:50, ng_sgm.m:20."""
import ctypes
import numpy as np
import pytest

from fsgm_amd import synth
from tests import mexharness as mh

pytestmark = pytest.mark.gpu


def test_calc_cost_sgm_gateway_as_epipolar_sgm_of_calls_it(gpu_lib, oracle):
    W, H, D = 96, 64, 64
    I1, I2 = synth.image_pair(W, H, D, seed=3)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=6)
    (bestD, minC), _ = mh.call("calc_cost_sgm", 2, I1, I2, D, 0.3, pd0, nd, off, 6, 64)
    rbd, rmc = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, 4)      # as shipped: 4 paths
    assert bestD.dtype == np.uint32 and bestD.shape == (H, W)
    np.testing.assert_array_equal(bestD, rbd)
    np.testing.assert_array_equal(minC, rmc)
    outs, _ = mh.call("calc_cost_sgm", 4, I1, I2, D, 0.3, pd0, nd, off, 6, 64)
    assert outs[2].dtype == np.uint8 and not outs[2].any() and outs[3].dtype == np.uint32 and not outs[3].any()
    (only,), _ = mh.call("calc_cost_sgm", 0, I1, I2, D, 0.3, pd0, nd, off, 6, 64)
    np.testing.assert_array_equal(only, rbd)
    # FSGM_EPI_FB_CHECK=1: the commented-out forward-backward check fills outputs 3 and 4
    import os
    os.environ["FSGM_EPI_FB_CHECK"] = "1"
    try:
        outs, _ = mh.call("calc_cost_sgm", 4, I1, I2, D, 0.3, pd0, nd, off, 6, 64)
    finally:
        del os.environ["FSGM_EPI_FB_CHECK"]
    Cv = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    idx, _ = oracle.epi_wta(oracle.epi_aggregate(Cv, 6, 64, 4), W, H, D, 1)
    conf, d2 = oracle.epi_fb_check(idx, pd0, nd, off, 0.3, D + 1)
    np.testing.assert_array_equal(outs[0], rbd)
    np.testing.assert_array_equal(outs[2], conf)
    np.testing.assert_array_equal(outs[3], d2)


def test_calc_pyd_cost_sgm_gateway_as_pyramidal_sgm_calls_it(gpu_lib, oracle):
    W, H = 72, 40
    I1, I2 = synth.image_pair(W, H, 16, seed=8)
    mv = synth.hint_map(W, H, "even", seed=1)
    (minIdx, minC, mvSub), printed = mh.call("calc_pyd_cost_sgm", 3, I1, I2, mv, 5, 5, 2, 1, 6, 32, 1, 2, 0)
    bd, mc, ms = oracle.calc_pyd_cost_sgm(I1, I2, mv, 5, 5, 2, 1, 6, 32, 1, 2, 0)
    assert printed == f"width: {W}, height: {H}, dMax: 121, winRadiusAgg: 2\n"   # calc_pyd_cost_sgm.cpp:491
    assert mvSub.shape == (2, H, W)
    np.testing.assert_array_equal(minIdx, bd)
    np.testing.assert_array_equal(minC, mc)
    np.testing.assert_array_equal(mvSub, ms)


def test_ng_gateways_as_ng_sgm_calls_them(gpu_lib, oracle):
    W, H = 36, 24
    I1, I2 = synth.image_pair(W, H, 16, seed=5)
    hints = np.zeros((2, H, W))                             # ng_sgm.m:17 passes zeros(row, col): one plane only ...
    (minC, flow), printed = mh.call("calc_pyd_cost_sgm_ng", 2, I1, I2, hints, 1, 2, 0, 6, 32)
    rmc, rfl = oracle.calc_pyd_cost_sgm_ng(I1, I2, hints, 1, 2, 0, 6, 32)
    assert "dMax: 81" in printed
    np.testing.assert_array_equal(minC, rmc)
    np.testing.assert_array_equal(flow, rfl)
    # the single-plane W x H hint ng_sgm.m really passes: mvHeight = H/2, lower half = y plane
    (minC1, flow1), _ = mh.call("calc_pyd_cost_sgm_ng", 2, I1, I2, hints[0], 1, 2, 0, 6, 32)
    r1mc, r1fl = oracle.calc_pyd_cost_sgm_ng(I1, I2, hints[0].reshape(2, H // 2, W), 1, 2, 0, 6, 32)
    np.testing.assert_array_equal(minC1, r1mc)
    np.testing.assert_array_equal(flow1, r1fl)

    ctypes.CDLL(None).srand(ctypes.c_uint(1))
    (minC, flow), printed = mh.call("calc_cost_sgm_ng", 2, I1, I2, hints[0], 1, 2, 0, 6, 32)   # ... which this MEX ignores
    rs = oracle.glibc_rand_stream(oracle.sgm_ng_rand_draws(W, H), seed=1)
    rmc, rfl = oracle.calc_cost_sgm_ng(I1, I2, 6, 32, rs)
    assert printed == "dMax : 108\n"
    np.testing.assert_array_equal(minC, rmc)
    np.testing.assert_array_equal(flow, rfl)


def test_fsgm_pyramidal_sgm_gateway_as_a_drop_in_pyramidal_sgm_would_call_it(gpu_lib, oracle):
    """[mv, minC, mvPyd1..3] = fsgm_pyramidal_sgm(permute(I0,[2 1 3]), permute(I1,[2 1 3]), 3) -- the
    argument values of test_psgm.m:33-34 (RGB pair, three levels)."""
    W, H = 85, 47
    g0, g1 = synth.image_pair(W, H, 12, seed=4)
    I0 = np.stack([g0, 255 - g0, g0 // 3 + 80])
    I1 = np.stack([g1, 255 - g1, g1 // 3 + 80])
    outs, _ = mh.call("fsgm_pyramidal_sgm", 5, I0, I1, 3)
    want_mv, want_minC, want_lv = oracle.pyramidal_sgm(I0, I1, 3)
    assert outs[0].shape == (2, H, W) and outs[0].dtype == np.float64 and outs[1].dtype == np.uint32
    np.testing.assert_array_equal(outs[0], want_mv)
    np.testing.assert_array_equal(outs[1], want_minC)
    for l in range(3):
        np.testing.assert_array_equal(outs[2 + l], want_lv[l])
    (only,), _ = mh.call("fsgm_pyramidal_sgm", 1, g0, g1)                       # gray pair, default numPyd = 5
    np.testing.assert_array_equal(only, oracle.pyramidal_sgm(g0, g1, 5)[0])


def test_fsgm_pyramidal_sgm_ng_gateway(gpu_lib, oracle):
    """[flow, minC, flowPyd1..3] = fsgm_pyramidal_sgm_ng(I0p, I1p, 3): the level loop around calc_pyd_cost_sgm_ng
    (BASELINE config 4), against the same composition of the oracle's functions."""
    W, H = 70, 41
    g0, g1 = synth.image_pair(W, H, 8, seed=5)
    I0 = np.stack([g0, 255 - g0, g0 // 3 + 80])
    I1 = np.stack([g1, 255 - g1, g1 // 3 + 80])
    outs, _ = mh.call("fsgm_pyramidal_sgm_ng", 5, I0, I1, 3)
    lv = [(I0, I1)]
    for _ in range(2):
        a, b = lv[-1]
        lv.append((np.stack([oracle.impyramid_reduce(c) for c in a]), np.stack([oracle.impyramid_reduce(c) for c in b])))
    gray = [(oracle.rgb2gray(a), oracle.rgb2gray(b)) for a, b in lv]
    mvPre = np.zeros((2,) + gray[-1][0].shape)
    want = {}
    for l in (3, 2, 1):
        mc, fl = oracle.calc_pyd_cost_sgm_ng(gray[l - 1][0], gray[l - 1][1], mvPre, 1, 2, 0, 6, 32)
        want[l] = fl
        mvPre = np.ascontiguousarray(2.0 * np.repeat(np.repeat(fl, 2, axis=1), 2, axis=2))
    assert outs[0].shape == (2, H, W) and outs[0].dtype == np.float64 and outs[1].dtype == np.uint32
    np.testing.assert_array_equal(outs[0], want[1])
    np.testing.assert_array_equal(outs[1], mc)
    for l in (1, 2, 3):
        np.testing.assert_array_equal(outs[1 + l], want[l])
