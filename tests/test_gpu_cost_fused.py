"""GPU parity tests for the fused cost stage (epi_cost.hip: raw census cost + 5x5 box mean in one kernel,
calc_cost_sgm.cpp:319-412): against the oracle, against the two-kernel form (FSGM_COST_FUSED=0), at the strip / segment /
border geometries the kernel cuts a frame into, and on sample positions planted within an ulp of every rounding boundary.
"""
import os

import numpy as np
import pytest

from fsgm_amd import synth, EpiPlan
from fsgm_amd._lib import STAGE_COST

pytestmark = pytest.mark.gpu


def _cost(W, H, D, frames, fused=True):
    """C of every frame through the plan (frames: list of (I1, I2, pd0, nd, off))."""
    old = os.environ.get("FSGM_COST_FUSED")
    os.environ["FSGM_COST_FUSED"] = "1" if fused else "0"
    try:
        with EpiPlan(W, H, D, len(frames), paths=8) as plan:
            plan.set_penalties(6, 64, 0.3)
            for f, fr in enumerate(frames):
                plan.upload(f, *fr)
            plan.run(STAGE_COST)
            return [plan.download_cost(f) for f in range(len(frames))]
    finally:
        if old is None:
            os.environ.pop("FSGM_COST_FUSED", None)
        else:
            os.environ["FSGM_COST_FUSED"] = old


def _frame(W, H, D, seed, kind="general"):
    I1, I2 = synth.image_pair(W, H, D, seed=seed)
    pd0, nd, off = synth.epi_maps(W, H, kind, seed=seed + 3)
    return I1, I2, pd0, nd, off


# strips of 60 output columns (64 raw): one column, one short of / exactly / one past a strip, several strips with a ragged
# last one; rows: fewer than the 5-row window, segment boundaries (16-row segments for a single frame)
@pytest.mark.parametrize("W,H,D", [
    (1, 1, 16), (2, 3, 16), (3, 2, 32), (59, 4, 16), (60, 5, 16), (61, 17, 32), (64, 16, 64), (119, 33, 16), (120, 15, 128),
    (121, 31, 128), (181, 48, 64), (70, 35, 256), (5, 70, 16), (300, 7, 16),
])
def test_fused_cost_matches_the_oracle(gpu_lib, oracle, W, H, D):
    fr = _frame(W, H, D, seed=W + 2 * H + D)
    want = oracle.epi_cost(fr[0], fr[1], D, 0.3, *fr[2:])
    got = _cost(W, H, D, [fr])[0]
    assert got.max() <= 24
    np.testing.assert_array_equal(got, want)


def test_fused_and_two_kernel_forms_agree_on_a_batch(gpu_lib, oracle):
    W, H, D = 131, 77, 128
    frames = [_frame(W, H, D, seed=40 + i, kind="general" if i % 2 else "axis") for i in range(3)]
    a = _cost(W, H, D, frames, fused=True)
    b = _cost(W, H, D, frames, fused=False)
    for i, fr in enumerate(frames):
        np.testing.assert_array_equal(a[i], b[i])
    np.testing.assert_array_equal(a[2], oracle.epi_cost(frames[2][0], frames[2][1], D, 0.3, *frames[2][2:]))


def test_fused_cost_long_segments(gpu_lib, oracle):
    """A batch large enough for the launcher to take its long row segments (>= 1024 workgroups: 2 segments of 75 rows here)."""
    W, H, D, B = 130, 150, 16, 180
    base = [_frame(W, H, D, seed=70 + i) for i in range(3)]
    frames = [base[i % 3] for i in range(B)]
    got = _cost(W, H, D, frames)
    for i in (0, 1, 2, 91, B - 1):
        np.testing.assert_array_equal(got[i], oracle.epi_cost(frames[i][0], frames[i][1], D, 0.3, *frames[i][2:]))


def _plant(target, ox):
    """Pd0 such that fl(fl(Pd0 - 1) + ox) == target exactly (searching the neighbouring doubles), NaN where none is found."""
    pd = (target - ox) + 1.0
    out = np.full_like(pd, np.nan)
    cand = pd.copy()
    for direction in (0.0, np.inf, -np.inf):
        c = cand.copy()
        for _ in range(6):
            ok = ((c - 1.0) + ox) == target
            out = np.where(np.isnan(out) & ok, c, out)
            if direction == 0.0:
                break
            c = np.nextafter(c, direction)
    return out


def test_sample_positions_within_an_ulp_of_every_half(gpu_lib, oracle):
    """For every pixel one disparity's sample coordinate is planted ON k + 0.5, one ulp below it or one ulp above it (both
    coordinates, both signs of the travelled offset, k from 0 to beyond the image), with non-trivial products off * vz * u
    in front of the final addition; plus pred(0.5) itself, where floor(v + 0.5) is wrong."""
    W, H, D = 97, 41, 32
    I1, I2 = synth.image_pair(W, H, D, seed=12)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=31)
    n = D + 1
    vz = np.array([(1.0 * d / n * 0.3) / (1 - (1.0 * d / n * 0.3)) for d in range(D)])
    yy, xx = np.mgrid[0:H, 0:W]
    dsel = (3 * xx + 5 * yy) % D                             # the disparity whose sample gets planted
    s = off * vz[dsel]
    for axis in (0, 1):
        ox = s * nd[axis]
        k = (xx * 7 + yy * 3 + axis) % (W + 6 if axis == 0 else H + 6)
        tgt = k + 0.5
        which = (xx + 2 * yy + axis) % 3                     # on the half, one ulp below, one ulp above
        tgt = np.where(which == 1, np.nextafter(tgt, -np.inf), np.where(which == 2, np.nextafter(tgt, np.inf), tgt))
        tgt[0, 0] = np.nextafter(0.5, 0.0)                   # pred(0.5): rounds to 0, floor(v + 0.5) says 1
        planted = _plant(tgt, ox)
        use = ~np.isnan(planted) & ((xx + yy + axis) % 2 == 0)           # the other coordinate of these pixels stays generic
        assert use.sum() > W * H // 4
        pd0[axis] = np.where(use, planted, pd0[axis])
    want = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    for fused in (True, False):
        got = _cost(W, H, D, [(I1, I2, pd0, nd, off)], fused=fused)[0]
        np.testing.assert_array_equal(got, want)


def test_fused_cost_rows_that_leave_the_fast_rounding(gpu_lib, oracle):
    """Rows with a lane whose sample positions may reach 2^31 (or are not finite) take the kernel's restated x86 conversion;
    their neighbours above and below stay on the fast path: all of them must match the oracle."""
    W, H, D = 75, 23, 64
    I1, I2, pd0, nd, off = _frame(W, H, D, seed=5)
    off[3, 4] = 1e12
    off[7, 70] = -1e12
    off[8, 0] = 1e300
    pd0[0, 9, 9] = np.inf
    pd0[1, 10, 10] = np.nan
    pd0[0, 12, 61] = 2147483648.5
    pd0[1, 12, 62] = 2147483647.5 + 1.0
    nd[0, 15, 30] = np.nan
    off[16, 33] = np.inf
    pd0[0, 22, 74] = -1e300
    want = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    got = _cost(W, H, D, [(I1, I2, pd0, nd, off)])[0]
    np.testing.assert_array_equal(got, want)


def test_rows_with_horizontal_epipolar_lines_take_the_x_only_offsets(gpu_lib, oracle):
    """Rows whose lanes all have uy == 0 (of either sign) skip the y coordinate; rows with a single lane that has not, rows with
    uy == 0 but a fractional start row (round(by) is not y), and a NaN direction stay exact."""
    W, H, D = 130, 40, 32
    I1, I2, pd0, nd, off = _frame(W, H, D, seed=21)
    nd[1, 0:30] = 0.0                                        # horizontal lines in rows 0..29
    nd[1, 3:6] = -0.0
    nd[0, 0:30] = np.where(nd[0, 0:30] >= 0, 1.0, -1.0)
    nd[1, 7, 64] = 1e-300                                    # one lane of a row that is not horizontal after all
    nd[1, 9, 129] = np.nan
    pd0[1, 11] += 0.5                                        # start rows on halves: round-half-away of by
    pd0[1, 12] -= 0.49999999999999994
    pd0[1, 13] = -3.0                                        # start row above the image: clamps to row 0
    pd0[1, 14] = H + 7.25                                    # below: clamps to the last row
    want = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    for fused in (True, False):
        got = _cost(W, H, D, [(I1, I2, pd0, nd, off)], fused=fused)[0]
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("W,H", [(16, 5), (61, 47), (64, 9), (130, 21), (23, 40)])
def test_four_pixel_census_kernel(gpu_lib, oracle, W, H):
    """census5x5_quad_kernel (taken by size in production: forced here) against the oracle and the one-pixel kernel: interior
    threads read 12 bytes a row as three dwords at any alignment, border threads replicate (common.cpp:17-18)."""
    import fsgm_amd
    img = synth.uniform_u8(W * 7 + H, (H, W))
    img[::3, ::5] = img[1, 1]                                # ties: nbr >= ctr
    os.environ["FSGM_CENSUS_QUAD"] = "1"
    try:
        quad = fsgm_amd.census(img)
    finally:
        os.environ["FSGM_CENSUS_QUAD"] = "0"
    try:
        one = fsgm_amd.census(img)
    finally:
        del os.environ["FSGM_CENSUS_QUAD"]
    np.testing.assert_array_equal(quad, oracle.census(img))
    np.testing.assert_array_equal(quad, one)
