"""GPU parity tests for the post-processing chain (test.m:45-50: speckle_filter.m, calc_disp_from_first.m,
forward_backward_check.m, scanline_in_fill.m, vzInd2Disp.m) through the C ABI vs the CPU oracle
(oracle/fsgm_oracle_post.cpp, which follows the originals' raster scans and flood fill).  Values are
copies / minima / IEEE expressions of the inputs: compared exactly, NaN positions included."""
import numpy as np
import pytest

import fsgm_amd
from fsgm_amd import synth, PostPlan

pytestmark = pytest.mark.gpu


def _same(a, b, msg=""):
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b), err_msg=msg)
    np.testing.assert_array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0), err_msg=msg)


@pytest.mark.parametrize("W,H,maxDiff,maxSize,inv", [(40, 30, 2, 100, 0.08), (37, 23, 0.5, 6, 0.08), (300, 200, 2, 100, 0.05),
                                                     (257, 65, 64, 257 * 65 / 10, 0.3), (9, 1, 2, 3, 0.1), (1, 9, 2, 3, 0.1),
                                                     (1242, 375, 2, 100, 0.08), (130, 70, 1.0, 1e9, 0.2), (130, 70, 1.0, 0, 0.2)])
def test_speckle_filter_bit_exact(gpu_lib, oracle, W, H, maxDiff, maxSize, inv):
    img = synth.vz_index_map(W, H, 64, seed=W + H, invalid=inv)
    out, labels = fsgm_amd.speckle_filter(img, maxDiff, maxSize)
    want, wlabels = oracle.speckle_filter(img, maxDiff, maxSize)
    _same(out, want)
    np.testing.assert_array_equal(labels, wlabels)


def test_speckle_filter_one_big_snake(gpu_lib, oracle):
    """A single region that winds through the whole image (worst case for union-find chains)."""
    H, W = 63, 96
    img = np.full((H, W), np.nan)
    for y in range(0, H, 2):
        img[y, :] = 1.0
        if y + 1 < H:
            img[y + 1, (W - 1) if (y // 2) % 2 == 0 else 0] = 1.0
    out, labels = fsgm_amd.speckle_filter(img, 2, 100)
    want, wlabels = oracle.speckle_filter(img, 2, 100)
    _same(out, want)
    np.testing.assert_array_equal(labels, wlabels)
    assert labels.max() == 1


@pytest.mark.parametrize("W,H,inv", [(24, 15, 0.35), (7, 5, 0.35), (9, 1, 0.3), (1, 6, 0.3), (700, 41, 0.5), (1242, 375, 0.2), (300, 7, 0.97)])
def test_scanline_in_fill_bit_exact(gpu_lib, oracle, W, H, inv):
    a = synth.vz_index_map(W, H, 64, seed=H, invalid=inv)
    _same(fsgm_amd.scanline_in_fill(a), oracle.scanline_in_fill(a))
    a[:] = np.nan
    assert np.isnan(fsgm_amd.scanline_in_fill(a)).all()


@pytest.mark.parametrize("W,H,kind", [(28, 19, "general"), (320, 240, "general"), (1242, 375, "axis")])
def test_disparity_functions_bit_exact(gpu_lib, oracle, W, H, kind):
    D, vMax = 64, 0.3
    D1 = synth.vz_index_map(W, H, D, seed=3)
    pd0, nd, off = synth.epi_maps(W, H, kind, seed=5)
    off = off / 8
    D2 = fsgm_amd.calc_disp_from_first(D1, pd0, nd, off, vMax, D + 1)
    _same(D2, oracle.calc_disp_from_first(D1, pd0, nd, off, vMax, D + 1))
    _same(fsgm_amd.forward_backward_check(D1, D2, pd0, nd, off, vMax, D + 1),
          oracle.forward_backward_check(D1, D2, pd0, nd, off, vMax, D + 1))
    _same(fsgm_amd.vzInd2Disp(D1, off, vMax, D + 1), oracle.vzind2disp(D1, off, vMax, D + 1))
    with pytest.raises(fsgm_amd.FsgmError):                  # the max-form of the scatter needs non-negative maps
        fsgm_amd.calc_disp_from_first(D1 - 100.0, pd0, nd, off, vMax, D + 1)


@pytest.mark.parametrize("W,H,D", [(45, 31, 32), (320, 240, 64), (1242, 375, 64)])
def test_postprocess_chain_bit_exact(gpu_lib, oracle, W, H, D):
    """test.m:45-50 in one device-resident call, on a map made like the real thing: the vz indices of
    calc_cost_sgm on a synthetic pair (bestD / 256 before the disparity conversion) with holes punched in."""
    vMax = 0.3
    D1 = synth.vz_index_map(W, H, D, seed=8)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=2)
    off = off / 8
    want = oracle.postprocess(D1, pd0, nd, off, vMax, D + 1, D)
    got = fsgm_amd.epi_postprocess(D1, pd0, nd, off, vMax, D + 1, D)
    for g, w, name in zip(got, want, ("filterD1", "filterD2", "filterdisparites")):
        _same(g, w, name)
    with PostPlan(W, H) as plan:                             # plan form, run twice (scratch must be re-initialised)
        plan.upload(D1, pd0, nd, off)
        for _ in range(2):
            plan.run(vMax, D + 1, D)
            for g, w, name in zip(plan.download(), want, ("filterD1", "filterD2", "filterdisparites")):
                _same(g, w, name)


def test_postprocess_after_calc_cost_sgm(gpu_lib, oracle):
    """End to end on this repo's own SGM output: bestD of calc_cost_sgm (x256 fixed point, before the
    disparity conversion) -> D1 = bestD/256 like epipolar_sgm_of.m:46 -> the chain."""
    W, H, D, vMax = 200, 120, 64, 0.3
    I1, I2 = synth.image_pair(W, H, D, seed=4)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=6)
    off = off / 8
    from fsgm_amd import EpiPlan
    with EpiPlan(W, H, D, 1, paths=8, vz_to_disp=0) as plan:
        plan.set_penalties(6, 64, vMax)
        plan.upload(0, I1, I2, pd0, nd, off)
        plan.run()
        bestD, _ = plan.download(0)
    D1 = bestD.astype(np.float64) / 256.0
    want = oracle.postprocess(D1, pd0, nd, off, vMax, D + 1, D)
    got = fsgm_amd.epi_postprocess(D1, pd0, nd, off, vMax, D + 1, D)
    for g, w in zip(got, want):
        _same(g, w)


@pytest.mark.parametrize("W,H,ch", [(27, 19, 3), (5, 3, 1), (1, 1, 2), (1242, 375, 3)])
def test_vmf_bit_exact(gpu_lib, oracle, W, H, ch):
    flow = (synth.uniform_f64(W, (ch, H, W)) - 0.5) * 40
    flow[-1] = 1.0
    np.testing.assert_array_equal(fsgm_amd.vmf(flow), oracle.vmf(flow))
