"""CPU tests of the pyramidal_sgm.m oracle (oracle/fsgm_oracle_pyramid.cpp, PARITY UNPINNED: the driver
calls MATLAB toolbox functions that are not part of the reference tree, see the file header).  The
oracle's integer forms of impyramid / rgb2gray / nearest up-sampling must agree with a second
restatement written as imresize's own contribution tables (tests/py_restatement.py), and the whole
driver with a second restatement in raster order."""
import numpy as np
import pytest

from fsgm_amd import synth
from tests import py_restatement as R


@pytest.mark.parametrize("W,H", [(16, 12), (17, 13), (5, 3), (2, 2), (1, 7), (9, 1), (1, 1), (64, 37)])
def test_impyramid_reduce_vs_contribution_tables(oracle, W, H):
    img = synth.uniform_u8(W * 31 + H, (H, W))
    got = oracle.impyramid_reduce(img)
    assert got.shape == ((H + 1) // 2, (W + 1) // 2)
    np.testing.assert_array_equal(got, R.impyramid_reduce(img))


def test_impyramid_reduce_known_answers(oracle):
    """Hand-checkable cases: a constant image stays constant; an impulse spreads with weights
    [1 4 6 4 1]/16 per axis, rounded after each axis, and only even input samples are centres."""
    np.testing.assert_array_equal(oracle.impyramid_reduce(np.full((7, 9), 93, np.uint8)), np.full((4, 5), 93, np.uint8))
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255                                     # centre of output sample (2, 2)
    out = oracle.impyramid_reduce(img).astype(int)
    col = [(255 * w + 8) // 16 for w in (1, 6, 1)]      # rows 1,2,3 see the impulse with weights 1,6,1 (taps 2i-2..2i+2)
    assert [out[1, 2], out[2, 2], out[3, 2]] == [(c * 6 + 8) // 16 for c in col]
    assert out[2, 1] == (col[1] * 1 + 8) // 16 and out[0].sum() == 0 and out[4].sum() == 0
    img = np.zeros((9, 9), np.uint8)
    img[4, 3] = 160                                     # odd column: weights 4 and 4 on output columns 1 and 2
    out = oracle.impyramid_reduce(img).astype(int)
    assert out[2, 1] == out[2, 2] == (((160 * 6 + 8) // 16) * 4 + 8) // 16


def test_rgb2gray_vs_second_restatement(oracle):
    rgb = synth.uniform_u8(77, (3, 23, 31))
    rgb[:, 0, :8] = np.array([[0, 255, 255, 0, 0, 255, 1, 254]] * 3)
    rgb[1, 0, 1:4] = (0, 255, 0)
    got = oracle.rgb2gray(rgb)
    np.testing.assert_array_equal(got, R.rgb2gray(rgb))
    assert got[0, 0] == 0 and got[0, 5] == 255          # the weights sum to one


def test_nearest_upsampling_is_duplication():
    a = np.arange(12.0).reshape(3, 4)
    np.testing.assert_array_equal(R.resize2_nearest(a), np.repeat(np.repeat(a, 2, axis=0), 2, axis=1))


@pytest.mark.parametrize("W,H,ch,numPyd,ver,hor", [(13, 9, 1, 2, 1, 2), (10, 7, 3, 2, 2, 1), (9, 6, 1, 3, 1, 1), (7, 5, 1, 1, 2, 2)])
def test_pyramidal_sgm_vs_second_restatement(oracle, W, H, ch, numPyd, ver, hor):
    I0, I1 = synth.image_pair(W, H, 6, seed=W + H)
    if ch == 3:
        I0 = np.stack([I0, np.roll(I0, 1, axis=1), 255 - I0])
        I1 = np.stack([I1, np.roll(I1, 1, axis=1), 255 - I1])
    mv, minC, lv = oracle.pyramidal_sgm(I0, I1, numPyd, 6, 32, 2, ver, hor, 1, 2, 0)
    rmv, rminC, rlv = R.pyramidal_sgm(I0, I1, numPyd, 6, 32, 2, ver, hor, 1, 2, 0)
    np.testing.assert_array_equal(mv, rmv)
    np.testing.assert_array_equal(minC, rminC)
    for a, b in zip(lv, rlv):
        np.testing.assert_array_equal(a, b)


def test_layers_regression_fixture(oracle):
    """Guards the oracle's pyramidal driver, post-processing chain and epipolar maps against accidental
    edits.  NOT reference-derived: produced by this oracle itself (tests/golden/make_oracle_fixtures.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_layers.npz"))
    g0, g1 = synth.image_pair(40, 26, 10, seed=21)
    R0 = np.stack([g0, 255 - g0, g0 // 3 + 80])
    R1 = np.stack([g1, 255 - g1, g1 // 3 + 80])
    mv, minC, lv = oracle.pyramidal_sgm(R0, R1, 3)
    np.testing.assert_array_equal(mv, g["pyr_mv"])
    np.testing.assert_array_equal(minC, g["pyr_minC"])
    np.testing.assert_array_equal(lv[1], g["pyr_lv2"])
    np.testing.assert_array_equal(lv[2], g["pyr_lv3"])
    D1 = synth.vz_index_map(44, 30, 32, seed=22)
    pd0, nd, off = synth.epi_maps(44, 30, "general", seed=23)
    for got, key in zip(oracle.postprocess(D1, pd0, nd, off / 8, 0.3, 33, 32), ("post_f1", "post_f2", "post_disp")):
        np.testing.assert_array_equal(np.isnan(got), np.isnan(g[key]))
        np.testing.assert_array_equal(np.nan_to_num(got, nan=-7.0), np.nan_to_num(g[key], nan=-7.0))
    F, Hm, epi, direction = synth.epi_geometry(36, 24, "contract")
    for got, key in zip(oracle.epipolar_maps(F, Hm, epi, direction, 36, 24), ("geo_Pd0", "geo_nd", "geo_off", "geo_rflow")):
        np.testing.assert_array_equal(got, g[key])
