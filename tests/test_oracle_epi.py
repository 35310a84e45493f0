"""CPU tests of the oracle for the calc_cost_sgm path (no GPU).

* census: pinned bit-for-bit against the reference's own common.cpp (oracle/_ref, built in the
  build container; the committed golden vector was generated from it by
  tests/golden/make_census_golden.py).
* everything else: PARITY UNPINNED w.r.t. the compiled reference (its MEX sources need MATLAB's
  mex.h).  Checked here against an independent second restatement (tests/py_restatement.py) and
  against structural properties of the algorithm.
"""
import os
import numpy as np
import pytest

from fsgm_amd import synth
from tests import py_restatement as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_census_golden_from_reference(oracle):
    g = np.load(os.path.join(GOLD, "census_ref_61x47.npz"))
    np.testing.assert_array_equal(oracle.census(g["img"]), g["cen"])


def test_census_matches_reference_build(oracle):
    if not oracle.ref_census_available():
        pytest.skip("oracle/_ref not built (reference tree absent on this box)")
    for (W, H, seed) in [(1, 1, 1), (2, 3, 2), (5, 5, 3), (64, 48, 4), (131, 7, 5), (3, 97, 6)]:
        img = synth.uniform_u8(seed, (H, W))
        np.testing.assert_array_equal(oracle.census(img), oracle.ref_census(img))
    flat = np.full((9, 11), 77, np.uint8)                       # ties: nbr >= ctr is true everywhere
    np.testing.assert_array_equal(oracle.census(flat), oracle.ref_census(flat))
    assert (oracle.census(flat) == 0x3FFFFFE).all()


def test_census_bit_layout(oracle):
    img = synth.uniform_u8(9, (20, 30))
    cen = oracle.census(img)
    assert (cen & 1 == 0).all()                                  # trailing shift, common.cpp:21
    assert ((cen >> 13) & 1 == 1).all()                          # centre tap compares equal
    assert (cen < (1 << 26)).all()
    np.testing.assert_array_equal(cen, R.census(img))


def test_box_mean_integer_form():
    """(u8)(1.0*sum/25 + 0.5) == (2*sum+25)//50 for every reachable sum (calc_cost_sgm.cpp:404)."""
    s = np.arange(0, 25 * 255 + 1)
    a = (1.0 * s / 25 + 0.5).astype(np.int64) & 0xFF
    b = ((2 * s + 25) // 50) & 0xFF
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("kind", ["axis", "general"])
def test_cost_volume_vs_second_restatement(oracle, kind):
    W, H, D = 14, 11, 8
    I1, I2 = synth.image_pair(W, H, D, seed=3)
    pd0, nd, off = synth.epi_maps(W, H, kind, seed=5)
    np.testing.assert_array_equal(oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off), R.calc_cost(I1, I2, D, 0.3, pd0, nd, off))


@pytest.mark.parametrize("P1,P2,cmax", [(6, 64, 24), (100, 200, 255), (0, 0, 255), (255, 255, 255), (7, 3, 60)])
@pytest.mark.parametrize("paths", [4, 8])
def test_aggregate_vs_second_restatement(oracle, P1, P2, cmax, paths):
    W, H, D = 9, 7, 6
    Cv = synth.cost_volume(W, H, D, seed=P1 + paths, cmax=cmax)
    got = oracle.epi_aggregate(Cv, P1, P2, paths)
    assert got[-1] == 0
    np.testing.assert_array_equal(got[:-1].reshape(H, W, D), R.sgm_raster(Cv, P1, P2, paths == 8))


def test_wta_and_vz_vs_second_restatement(oracle):
    W, H, D = 10, 8, 8
    Cv = synth.cost_volume(W, H, D, seed=4, cmax=24)
    Cv[1, 2, :] = 24; Cv[1, 2, D - 1] = 0                        # best == D-1: reads the next pixel's d=0
    Cv[H - 1, W - 1, :] = 24; Cv[H - 1, W - 1, D - 1] = 0        # last pixel: one past the array (defined 0)
    Cv[3, 3, :] = 24; Cv[3, 3, 1] = 0                            # best == 1 is never refined (:293)
    S = oracle.epi_aggregate(Cv, 6, 64, 8)
    _, _, off = synth.epi_maps(W, H, "general", seed=8)
    bd, mc = oracle.epi_wta(S, W, H, D, 1)
    rbd, rmc = R.wta(S[:-1].reshape(H, W, D))
    np.testing.assert_array_equal(mc, rmc)
    np.testing.assert_array_equal(bd, rbd)
    assert bd[3, 3] == 256 and bd[1, 2] != (D - 1) * 256
    np.testing.assert_array_equal(oracle.epi_vz_to_disp(bd, off, 0.3, D + 1), R.vz_to_disp(bd, off, 0.3, D + 1))
    bd0, _ = oracle.epi_wta(S, W, H, D, 0)
    np.testing.assert_array_equal(bd0, R.wta(S[:-1].reshape(H, W, D), False)[0])


def test_path_start_rule_and_bounds(oracle):
    """First pixel of a path stores L=C with minimum entry 0 (calc_cost_sgm.cpp:153-154), so on a
    1-row image the two vertical paths contribute exactly C each, and without wrap C <= L <= C+P2."""
    W, D = 17, 16
    Cv = synth.cost_volume(W, 1, D, seed=2, cmax=24)
    S = oracle.epi_aggregate(Cv, 6, 64, 4)[:-1].reshape(1, W, D)
    horiz = S.astype(np.int64) - 2 * Cv
    assert (horiz >= 2 * Cv.astype(np.int64)).all() and (horiz <= 2 * (Cv.astype(np.int64) + 64)).all()
    np.testing.assert_array_equal(horiz[0, 0], Cv[0, 0].astype(np.int64) + (S[0, 0] - 3 * Cv[0, 0].astype(np.int64)))


def test_mirror_symmetry(oracle):
    """Pass 1 is the point mirror of pass 0 (calc_cost_sgm.cpp:115-123)."""
    W, H, D = 23, 13, 16
    Cv = synth.cost_volume(W, H, D, seed=6, cmax=24)
    for paths in (4, 8):
        S0 = oracle.epi_aggregate(Cv, 6, 64, paths)[:-1].reshape(H, W, D)
        S1 = oracle.epi_aggregate(np.ascontiguousarray(Cv[::-1, ::-1]), 6, 64, paths)[:-1].reshape(H, W, D)
        np.testing.assert_array_equal(S1[::-1, ::-1], S0)


def test_whole_mex_equals_stage_composition(oracle):
    W, H, D = 40, 30, 16
    I1, I2 = synth.image_pair(W, H, D, seed=1)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=2)
    bd, mc, Cv, S = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, 8, want_volumes=True)
    np.testing.assert_array_equal(Cv, oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off))
    Sf = oracle.epi_aggregate(Cv, 6, 64, 8)
    np.testing.assert_array_equal(S, Sf[:-1].reshape(H, W, D))
    b2, m2 = oracle.epi_wta(Sf, W, H, D, 1)
    np.testing.assert_array_equal(mc, m2)
    np.testing.assert_array_equal(bd, oracle.epi_vz_to_disp(b2, off, 0.3, D + 1))


def test_oracle_regression_fixture(oracle):
    """Guards the oracle against accidental edits.  NOT a reference-derived vector: it was produced
    by this oracle itself (tests/golden/make_oracle_fixtures.py)."""
    g = np.load(os.path.join(GOLD, "oracle_epi_48x36x16.npz"))
    W, H, D = 48, 36, 16
    I1, I2 = synth.image_pair(W, H, D, seed=11)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=12)
    for paths in (4, 8):
        bd, mc = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths)
        np.testing.assert_array_equal(bd, g[f"bestD{paths}"])
        np.testing.assert_array_equal(mc, g[f"minC{paths}"])


@pytest.mark.parametrize("kind", ["axis", "general"])
def test_fb_check_vs_second_restatement(oracle, kind):
    """forward_backward_check / calc_disp_from_first (calc_cost_sgm.cpp:429-536, dead code in the shipped MEX)."""
    W, H, D = 20, 14, 16
    I1, I2 = synth.image_pair(W, H, D, seed=4)
    pd0, nd, off = synth.epi_maps(W, H, kind, seed=9)
    bd, _, _, S = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, 8, want_volumes=True)
    idx, _ = oracle.epi_wta(np.concatenate([S.reshape(-1), [0]]).astype(np.uint32), W, H, D, 1)
    conf, d2 = oracle.epi_fb_check(idx, pd0, nd, off, 0.3, D + 1)
    rconf, rd2 = R.fb_check(idx, pd0, nd, off, 0.3, D + 1)
    np.testing.assert_array_equal(d2, rd2)
    np.testing.assert_array_equal(conf, rconf)
