"""CPU tests of the epipolar driver's dense half (oracle/fsgm_oracle_geometry.cpp: rotation_motion.m,
epipolar_geometry.m:99-115; PARITY UNPINNED -- MATLAB scripts): oracle vs a whole-array numpy
restatement, plus properties the maps must have by construction."""
import numpy as np
import pytest

from fsgm_amd import synth
from tests import py_restatement as R


@pytest.mark.parametrize("W,H,kind", [(64, 48, "forward"), (61, 37, "contract"), (7, 1, "forward"), (1, 5, "contract")])
def test_epipolar_maps_vs_numpy(oracle, W, H, kind):
    F, Hm, epi, direction = synth.epi_geometry(max(W, 16), max(H, 16), kind)
    got = oracle.epipolar_maps(F, Hm, epi, direction, W, H)
    want = R.epipolar_maps(F, Hm, epi, direction, W, H)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


def test_epipolar_maps_properties(oracle):
    W, H = 96, 64
    F, Hm, epi, direction = synth.epi_geometry(W, H, "forward")
    Pd0, nd, off, rflow = oracle.epipolar_maps(F, Hm, epi, direction, W, H)
    yy, xx = np.mgrid[1:H + 1, 1:W + 1].astype(np.float64)
    np.testing.assert_array_equal(Pd0[0], xx + rflow[0])                       # PrefD0 = P + Rflow (:106)
    np.testing.assert_allclose(nd[0] ** 2 + nd[1] ** 2, 1.0, rtol=0, atol=1e-12)   # unit directions (:113)
    np.testing.assert_allclose(Pd0[0] - epi[0], off * nd[0], atol=1e-9)        # direction = (Pd0 - e') / offset
    c = oracle.epipolar_maps(F, Hm, epi, 1, W, H)                              # contraction: directions negated, rest equal
    np.testing.assert_array_equal(c[1], -nd)
    np.testing.assert_array_equal(c[2], off)
    # every zero-disparity point lies on its pixel's epipolar line: l2 . (Pd0 - 1, 1) ~ 0 (rotation_motion.m:27-28)
    x0, y0 = xx - 1, yy - 1
    l = [F[i, 0] * x0 + F[i, 1] * y0 + F[i, 2] for i in range(3)]
    nf = np.sqrt(l[0] ** 2 + l[1] ** 2)
    resid = (l[0] * (Pd0[0] - 1) + l[1] * (Pd0[1] - 1) + l[2]) / nf
    assert np.abs(resid).max() < 1e-9
