"""GPU parity tests for the dense half of the epipolar driver (epipolar_geometry.m:99-115,
rotation_motion.m, epipolar_sgm_of.m:33-51) through the C ABI vs the CPU oracle: maps and flows are
fp64 expressions evaluated in the same order without FMA -- compared exactly."""
import numpy as np
import pytest

import fsgm_amd
from fsgm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,kind", [(64, 48, "forward"), (61, 37, "contract"), (1242, 375, "forward"), (7, 1, "forward"), (1, 5, "contract")])
def test_epipolar_maps_bit_exact(gpu_lib, oracle, W, H, kind):
    F, Hm, epi, direction = synth.epi_geometry(max(W, 16), max(H, 16), kind)
    got = fsgm_amd.epipolar_maps(F, Hm, epi, direction, W, H)
    want = oracle.epipolar_maps(F, Hm, epi, direction, W, H)
    for g, w, name in zip(got, want, ("Pd0", "normlizeDirection", "Offset", "Rflow")):
        np.testing.assert_array_equal(g, w, err_msg=name)


@pytest.mark.parametrize("W,H,ch,kind,paths,dMax", [(96, 64, 1, "forward", 4, 64), (85, 47, 3, "contract", 4, 32), (160, 90, 3, "forward", 8, 64)])
def test_epipolar_sgm_of_bit_exact(gpu_lib, oracle, W, H, ch, kind, paths, dMax):
    """[flow, minC] = epipolar_sgm_of(I0, I1, K, dMax, vMax) with the sparse geometry given: images up,
    flow down; the maps never leave the device."""
    g0, g1 = synth.image_pair(W, H, dMax, seed=W)
    I0, I1 = (g0, g1) if ch == 1 else (np.stack([g0, 255 - g0, g0 // 3 + 80]), np.stack([g1, 255 - g1, g1 // 3 + 80]))
    F, Hm, epi, direction = synth.epi_geometry(W, H, kind)
    want_flow, want_minC = oracle.epipolar_sgm_of(I0, I1, F, Hm, epi, direction, dMax, 0.3, paths)
    flow, minC = fsgm_amd.epipolar_sgm_of(I0, I1, F, Hm, epi, direction, dMax, 0.3, paths=paths)
    np.testing.assert_array_equal(minC, want_minC)
    np.testing.assert_array_equal(flow, want_flow)
    assert (flow[2] == 1).all() and np.abs(flow[:2]).max() > 0


def test_epipolar_sgm_of_equals_the_separate_calls(gpu_lib):
    """The driver is calc_cost_sgm on the maps of epipolar_maps, then disparity * direction + rotation flow."""
    W, H, D = 120, 70, 64
    I0, I1 = synth.image_pair(W, H, D, seed=5)
    F, Hm, epi, direction = synth.epi_geometry(W, H, "forward")
    Pd0, nd, off, rflow = fsgm_amd.epipolar_maps(F, Hm, epi, direction, W, H)
    bestD, minC = fsgm_amd.calc_cost_sgm(I0, I1, D, 0.3, Pd0, nd, off, 6, 64)
    flow, minC2 = fsgm_amd.epipolar_sgm_of(I0, I1, F, Hm, epi, direction, D, 0.3)
    np.testing.assert_array_equal(minC2, minC)
    np.testing.assert_array_equal(flow[:2], (bestD.astype(np.float64) / 256.0) * nd + rflow)
