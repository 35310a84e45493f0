"""CPU tests of the oracle for calc_pyd_cost_sgm and calc_pyd_cost_sgm_ng (PARITY UNPINNED w.r.t.
the compiled reference): the C++ oracle (one recurrence per direction) must agree
bit for bit with an independent Python restatement written in the reference's raster order
(tests/py_restatement.py), on fractional hints, hint maps larger/smaller than the image, wrapping
penalties, adaptive P2 and 1/2/3 passes."""
import numpy as np
import pytest

from fsgm_amd import synth
from tests import py_restatement as R


@pytest.mark.parametrize("kind,mvpad", [("zero", 0), ("even", 0), ("general", 2)])
def test_pyd_cost_vs_second_restatement(oracle, kind, mvpad):
    W, H = 11, 9
    I1, I2 = synth.image_pair(W, H, 16, seed=3)
    mv = synth.hint_map(W + mvpad, H + mvpad, kind, seed=4, amp=3.0)
    c1, c2 = oracle.census(I1), oracle.census(I2)
    np.testing.assert_array_equal(oracle.pyd_cost(c1, c2, mv, 2, 2, 1), R.pyd_cost(c1, c2, mv, 2, 2, 1))
    np.testing.assert_array_equal(oracle.pyd_cost(c1, c2, mv, 1, 1, 2), R.pyd_cost(c1, c2, mv, 1, 1, 2))


@pytest.mark.parametrize("P1,P2,cmax,diag,passes,adaptive,kind", [
    (6, 32, 24, 1, 2, 0, "general"), (6, 32, 24, 0, 2, 1, "even"), (100, 200, 255, 1, 2, 1, "general"),
    (6, 32, 24, 1, 1, 0, "general"), (6, 32, 24, 1, 3, 0, "zero"),
])
def test_pyd_aggregate_and_wta_vs_second_restatement(oracle, P1, P2, cmax, diag, passes, adaptive, kind):
    W, H, Sx, Sy = 7, 6, 3, 5
    I1 = synth.uniform_u8(5, (H, W))
    mv = synth.hint_map(W + 1, H + 2, kind, seed=6, amp=2.5)
    Cv = synth.cost_volume(W, H, Sx * Sy, seed=P1, cmax=cmax)
    S = oracle.pyd_aggregate(I1, Cv, mv, Sx, Sy, P1, P2, diag, passes, adaptive)
    np.testing.assert_array_equal(S, R.pyd_sgm2d(I1, Cv, mv, Sx, Sy, P1, P2, diag, passes, adaptive))
    for sub in (0, 1):
        bd, mc, ms = oracle.pyd_wta(S, Sx, Sy, sub)
        rbd, rmc, rms = R.pyd_wta(S, Sx, Sy, sub)
        np.testing.assert_array_equal(bd, rbd)
        np.testing.assert_array_equal(mc, rmc)
        np.testing.assert_array_equal(ms, rms)


@pytest.mark.parametrize("P1,P2,agg,kind,mvshape", [(6, 32, 2, "general", (8, 6)), (90, 120, 5, "general", (5, 4)),
                                                    (6, 32, 3, "zero", (6, 5))])
def test_ng_vs_second_restatement(oracle, P1, P2, agg, kind, mvshape):
    W, H = 6, 5
    I1, I2 = synth.image_pair(W, H, 16, seed=9)
    mv = synth.hint_map(mvshape[0], mvshape[1], kind, seed=2, amp=5.0)
    minC, flow, Cc, S = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, agg, 0, P1, P2, want_volumes=True)
    rC = R.ng_cost(oracle.census(I1), oracle.census(I2), mv, agg // 2, 1)
    np.testing.assert_array_equal(Cc["mvx"], rC[..., 0])
    np.testing.assert_array_equal(Cc["mvy"], rC[..., 1])
    np.testing.assert_array_equal(Cc["cost"], rC[..., 2])
    rS, rminC, rflow = R.ng_sgm2d(rC, P1, P2)
    np.testing.assert_array_equal(S, rS)
    np.testing.assert_array_equal(minC, rminC)
    np.testing.assert_array_equal(flow, rflow)


@pytest.mark.parametrize("W,H,P1,P2", [(5, 4, 6, 32), (4, 5, 100, 200), (6, 1, 6, 32), (1, 4, 6, 32)])
def test_sgm_ng_vs_second_restatement(oracle, W, H, P1, P2):
    """calc_cost_sgm_ng.cpp: hints from the buffers about to be overwritten, top-N insertion, rand()."""
    I1, I2 = synth.image_pair(W, H, 16, seed=W * 3 + H)
    I1 = (I1.astype(np.int32) * 7 % 256).astype(np.uint8)
    rs = synth.splitmix64(7, oracle.sgm_ng_rand_draws(W, H)) % np.uint64(1 << 31)      # any non-negative stream
    rs = rs.astype(np.int32)
    mc, fl = oracle.calc_cost_sgm_ng(I1, I2, P1, P2, rs)
    rmc, rfl = R.otf(I1, I2, P1, P2, rs)
    np.testing.assert_array_equal(mc, rmc)
    np.testing.assert_array_equal(fl, rfl)
