"""CPU tests of the post-processing oracle (oracle/fsgm_oracle_post.cpp: speckle_filter.m,
calc_disp_from_first.m, forward_backward_check.m, scanline_in_fill.m, vzInd2Disp.m; PARITY UNPINNED --
MATLAB scripts, no MATLAB here, no stored outputs in the reference).  The oracle follows the originals'
raster scans and FIFO flood fill; it must agree with second restatements (tests/py_restatement.py), of
which the speckle filter uses a different algorithm (label propagation to a fixed point)."""
import numpy as np
import pytest

from fsgm_amd import synth
from tests import py_restatement as R


def _same(a, b):
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


@pytest.mark.parametrize("W,H,maxDiff,maxSize", [(40, 30, 2, 100), (37, 23, 0.5, 6), (16, 9, 64, 14.4), (9, 1, 2, 3), (1, 9, 2, 3)])
def test_speckle_filter_vs_label_propagation(oracle, W, H, maxDiff, maxSize):
    img = synth.vz_index_map(W, H, 32, seed=W)
    out, labels = oracle.speckle_filter(img, maxDiff, maxSize)
    rout, rlabels = R.speckle_filter(img, maxDiff, maxSize)
    _same(out, rout)
    np.testing.assert_array_equal(labels, rlabels)


def test_speckle_filter_known_answer(oracle):
    """Hand-checkable: a 2x2 blob of 5s in a field of 1s; a lone 9; threshold 4 removes the 9 only,
    threshold 5 the blob too; the strict < of the size test keeps a region of exactly maxSpeckleSize."""
    img = np.ones((6, 7))
    img[1:3, 1:3] = 5.0
    img[4, 5] = 9.0
    img[0, 6] = np.nan
    out, labels = oracle.speckle_filter(img, 2, 4)
    assert np.isnan(out[4, 5]) and not np.isnan(out[1:3, 1:3]).any() and np.isnan(out[0, 6])
    assert labels[0, 0] == 1 and labels[1, 1] == 2 and labels[4, 5] == 3 and labels[0, 6] == 0
    out, _ = oracle.speckle_filter(img, 2, 5)
    assert np.isnan(out[1:3, 1:3]).all() and np.isnan(out).sum() == 4 + 1 + 1
    out, _ = oracle.speckle_filter(img, 4.5, 5)          # |5-1| < 4.5: the blob joins the field; the 9 stays apart (|9-1| = 8)
    assert not np.isnan(out[1:3, 1:3]).any() and np.isnan(out[4, 5])


@pytest.mark.parametrize("W,H", [(24, 15), (7, 5), (9, 1), (1, 6), (3, 3)])
def test_scanline_in_fill_vs_literal_loops(oracle, W, H):
    a = synth.vz_index_map(W, H, 32, seed=H, invalid=0.35)
    _same(oracle.scanline_in_fill(a), R.scanline_in_fill(a))
    a[:] = np.nan                                        # nothing valid: stays NaN
    assert np.isnan(oracle.scanline_in_fill(a)).all()


def test_scanline_in_fill_known_answer(oracle):
    n = np.nan
    a = np.array([[n, 3.0, n, n, 1.0, n], [n, n, n, n, n, n], [2.0, n, 5.0, n, n, 4.0]])
    want = np.array([[3.0, 3.0, 1.0, 1.0, 1.0, 1.0], [n, n, n, n, n, n], [2.0, 2.0, 5.0, 4.0, 4.0, 4.0]])
    _same(oracle.scanline_in_fill(a), want)              # the all-NaN middle row is not reachable from top or bottom
    b = a[[1, 0, 2]]
    _same(oracle.scanline_in_fill(b)[0], want[0])        # ... but an all-NaN first row is (extrapolation to the top)


@pytest.mark.parametrize("W,H", [(28, 19), (11, 6)])
def test_disparity_functions_vs_literal_loops(oracle, W, H):
    D, vMax = 32, 0.3
    D1 = synth.vz_index_map(W, H, D, seed=3)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=5)
    off = off / 8                                        # targets mostly inside the image
    D2 = oracle.calc_disp_from_first(D1, pd0, nd, off, vMax, D + 1)
    _same(D2, R.calc_disp_from_first(D1, pd0, nd, off, vMax, D + 1))
    assert (D2 == -1).any() and (D2 >= 0).any()
    chk = oracle.forward_backward_check(D1, D2, pd0, nd, off, vMax, D + 1)
    _same(chk, R.forward_backward_check(D1, D2, pd0, nd, off, vMax, D + 1))
    assert np.isnan(chk).sum() > np.isnan(D1).sum() and (~np.isnan(chk)).any()
    _same(oracle.vzind2disp(D1, off, vMax, D + 1), R.vzInd2Disp(D1, off, vMax, D + 1))


def test_postprocess_chain_is_the_composition(oracle):
    W, H, D, vMax = 45, 31, 32, 0.3
    D1 = synth.vz_index_map(W, H, D, seed=8)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=2)
    off = off / 8
    f1, f2, disp = oracle.postprocess(D1, pd0, nd, off, vMax, D + 1, D)
    a, _ = R.speckle_filter(D1, 2, 100)                                          # test.m:45
    d2 = R.calc_disp_from_first(a, pd0, nd, off, vMax, D + 1)                    # :46
    b = R.forward_backward_check(a, d2, pd0, nd, off, vMax, D + 1)               # :47
    c, _ = R.speckle_filter(b, D, H * W / 10)                                    # :48
    e = R.scanline_in_fill(c)                                                    # :49
    _same(f2, d2)
    _same(f1, e)
    _same(disp, R.vzInd2Disp(e, off, vMax, D + 1))                               # :50


def test_vmf_vs_scipy_median_filter(oracle):
    """vmf.m = medfilt2 5x5 per channel with zero padding: scipy's median_filter(mode='constant') is the
    same definition (independent implementation)."""
    from scipy.ndimage import median_filter
    flow = (synth.uniform_f64(3, (3, 19, 27)) - 0.5) * 40
    flow[2] = 1.0
    got = oracle.vmf(flow)
    for c in range(3):
        np.testing.assert_array_equal(got[c], median_filter(flow[c], size=5, mode="constant", cval=0.0))
    assert got[2, 0, 0] == 0.0 and got[2, 9, 9] == 1.0      # corner: 16 of 25 window cells are padding zeros
