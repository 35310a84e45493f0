"""BASELINE.json's configurations at the sizes they name, on the GPU, against the CPU oracle.

config 4: "Pyramidal 2D-flow path (calc_pyd_cost_sgm_ng), 1242x375, 3-level pyramid, 1 GPU"
          calc_pyd_cost_sgm_ng.cpp:101-306 (sgm2d), :370-446 (calc_cost), :39-78 (sgm_step); level loop
          pyramidal_sgm.m:24-76.
config 5: "Batch of 8 KITTI-size pairs" -- its shape on one GPU (calc_cost_sgm.cpp:539-598, 8 paths).
census  : the device census kernel against the one artefact the reference's own sources pin here
          (tests/golden/census_ref_61x47.npz, generated from common.cpp:3-27).

The oracle needs ~30 s per 1242x375 neighbour-guided call (single-threaded, like the reference).
"""
import os
import numpy as np
import pytest

from fsgm_amd import synth, EpiPlan, calc_cost_sgm, calc_cost_sgm_batch, calc_pyd_cost_sgm, calc_pyd_cost_sgm_ng, census
from fsgm_amd._lib import STAGE_COST

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------------------------------- census
def test_census_kernel_against_the_reference_golden(gpu_lib):
    g = np.load(os.path.join(GOLD, "census_ref_61x47.npz"))
    np.testing.assert_array_equal(census(g["img"]), g["cen"])


@pytest.mark.parametrize("W,H", [(1, 1), (2, 3), (3, 2), (4, 4), (5, 5), (1, 40), (40, 1), (61, 47), (257, 33), (1242, 375)])
def test_census_kernel_border_shapes(gpu_lib, oracle, W, H):
    img = synth.uniform_u8(W * 31 + H, (H, W))
    np.testing.assert_array_equal(census(img), oracle.census(img))


def test_census_kernel_flat_image_ties(gpu_lib, oracle):
    flat = np.full((9, 11), 77, np.uint8)                       # ties: nbr >= ctr is true everywhere (common.cpp:19)
    got = census(flat)
    assert (got == 0x3FFFFFE).all()
    np.testing.assert_array_equal(got, oracle.census(flat))


def test_census_tap_of_the_plan(gpu_lib, oracle):
    """What FSGM_STAGE_COST leaves in the plan's census buffers = census() of both images."""
    g = np.load(os.path.join(GOLD, "census_ref_61x47.npz"))
    img = g["img"]
    H, W = img.shape
    I2 = np.ascontiguousarray(img[::-1, ::-1])
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=2)
    with EpiPlan(W, H, 16, 1) as plan:
        plan.upload(0, img, I2, pd0, nd, off)
        plan.run(STAGE_COST)
        c1, c2 = plan.download_census(0)
    np.testing.assert_array_equal(c1, g["cen"])                 # reference-derived
    np.testing.assert_array_equal(c2, oracle.census(I2))


# ------------------------------------------------------------------------------------------- config 4
def _smooth_hints(W, H, seed, big=False):
    """A smooth integer flow field (what a coarser level hands down): the 9 hints of a pixel, sampled 8 pixels
    apart, mostly repeat one another.  big=True adds regions whose vectors approach and cross the +-0x3FF0 range of
    the packed-key matcher (ng_kernels.hip) -- costs there are the out-of-image constant, the matcher is what runs."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    mv = np.stack([np.round(6.0 * np.sin(x / 90.0 + seed) + 3.0 * np.cos(y / 40.0)),
                   np.round(4.0 * np.sin(y / 70.0 - seed) + 2.0 * np.cos(x / 55.0))])
    mv[:, H // 3:H // 3 + 20, W // 4:W // 4 + 60] += 0.5        # fractional patch: equal vectors with different costs near borders
    if big:
        mv[0, 40:80, 100:300] = 16360.0                          # just inside 0x3FF0 = 16368
        mv[1, 40:80, 100:300] = -16365.0
        mv[0, 200:240, 600:900] = 16366.0                        # candidates 16365..16367 inside, hints of neighbours cross
        mv[1, 200:230, 700:800] = 16370.0                        # beyond the range: the exact comparison must take over
        mv[0, 300:330, 20:120] = -20000.0
    return np.ascontiguousarray(mv)


def test_config4_single_level_full_size(gpu_lib, oracle):
    """One calc_pyd_cost_sgm_ng call at 1242x375, 81 candidates, with every variant of the aggregation kernels the
    library can select for it (FSGM_NG_SPLIT parts, repeat removal on/off, list / grid form of the matcher): S, minC and the
    flow, all pixels."""
    W, H = 1242, 375
    I1, I2 = synth.image_pair(W, H, 16, seed=41)
    mv = _smooth_hints(W, H, 1, big=True)
    mc, fl, _, S = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 1, 6, 32, want_volumes=True)
    saved = {k: os.environ.get(k) for k in ("FSGM_NG_SPLIT", "FSGM_NG_DEDUPE", "FSGM_NG_GRID", "FSGM_NG_COMPACT")}
    try:
        # (matcher parts, repeat removal, grid form, compact kernel): "" = picked on the device -- these hints hold regions
        # outside the packed key's range, so the compact kernel (kept entries only) steps aside there and the split / list /
        # grid kernels run; "1" / "0" = forced / taken out of the set
        variants = [("2", "1", "", ""), ("1", "1", "", ""), ("2", "0", "", ""), ("1", "0", "", ""), ("3", "1", "", ""),
                    ("2", "1", "1", ""), ("1", "1", "0", ""), ("2", "1", "", "0"), ("1", "1", "", "0")]
        for split, dedupe, grid, compact in variants:
            os.environ["FSGM_NG_SPLIT"], os.environ["FSGM_NG_DEDUPE"] = split, dedupe
            for k, v in (("FSGM_NG_GRID", grid), ("FSGM_NG_COMPACT", compact)):
                if v:
                    os.environ[k] = v
                else:
                    os.environ.pop(k, None)
            gmc, gfl, gS = calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 1, 6, 32, return_sum=True)
            tag = f"split {split} dedupe {dedupe} grid {grid!r} compact {compact!r}"
            np.testing.assert_array_equal(gS, S, err_msg=tag)
            np.testing.assert_array_equal(gmc, mc, err_msg=tag)
            np.testing.assert_array_equal(gfl, fl, err_msg=tag)
        # the same frame without the out-of-range regions: every list fits, the compact kernel is what runs
        os.environ.pop("FSGM_NG_GRID", None); os.environ.pop("FSGM_NG_COMPACT", None)
        mv2 = _smooth_hints(W, H, 1, big=False)
        mc2, fl2, _, S2 = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv2, 1, 2, 1, 6, 32, want_volumes=True)
        gmc, gfl, gS = calc_pyd_cost_sgm_ng(I1, I2, mv2, 1, 2, 1, 6, 32, return_sum=True)
        np.testing.assert_array_equal(gS, S2, err_msg="compact")
        np.testing.assert_array_equal(gmc, mc2, err_msg="compact")
        np.testing.assert_array_equal(gfl, fl2, err_msg="compact")
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_config4_three_level_pyramid_full_size(gpu_lib, oracle):
    """BASELINE config 4 as named: 1242x375 RGB pair, 3 levels (1242x375 / 621x188 / 311x94), calc_pyd_cost_sgm_ng per
    level, every level's flow and the finest level's minC against the same composition of the oracle's functions."""
    from fsgm_amd import pyramidal_sgm_ng
    W, H = 1242, 375
    g0, g1 = synth.image_pair(W, H, 16, seed=2)
    I0 = np.stack([g0, 255 - g0, g0 // 2 + 40])
    I1 = np.stack([g1, 255 - g1, g1 // 2 + 40])
    flow, flows, minC = pyramidal_sgm_ng(I0, I1, 3)
    lv = [(I0, I1)]
    for _ in range(2):
        a, b = lv[-1]
        lv.append((np.stack([oracle.impyramid_reduce(c) for c in a]), np.stack([oracle.impyramid_reduce(c) for c in b])))
    gray = [(oracle.rgb2gray(a), oracle.rgb2gray(b)) for a, b in lv]
    assert [g[0].shape for g in gray] == [(375, 1242), (188, 621), (94, 311)]
    mvPre = np.zeros((2, 94, 311))
    for l in (3, 2, 1):
        mc, fl = oracle.calc_pyd_cost_sgm_ng(gray[l - 1][0], gray[l - 1][1], mvPre, 1, 2, 0, 6, 32)
        np.testing.assert_array_equal(flows[3 - l], fl, err_msg=f"level {l}")
        mvPre = np.ascontiguousarray(2.0 * np.repeat(np.repeat(fl, 2, axis=1), 2, axis=2))
    np.testing.assert_array_equal(minC, mc)
    np.testing.assert_array_equal(flow, fl)
    assert np.abs(flow).max() > 2                               # the hints did travel down the pyramid


# ------------------------------------------------------------------------------------------- config 5
def test_config5_batch_of_8_kitti_pairs(gpu_lib, oracle):
    """8 distinct 1242x375x128 pairs with 8 paths: fsgm_calc_cost_sgm_batch_host (a batch of 8 takes the parallel sweeps)
    with frames 0 and 7 against the oracle and every frame against a single-frame call (the line kernels); then the same
    8 pairs resident in a plan through the full sweep pipeline, every frame against the batch call."""
    W, H, D, B = 1242, 375, 128, 8
    frames = []
    for s in range(B):
        I1, I2 = synth.image_pair(W, H, D, seed=500 + s)
        pd0, nd, off = synth.epi_maps(W, H, "general" if s % 2 else "axis", seed=600 + s)
        frames.append((I1, I2, pd0, nd, off))
    res = calc_cost_sgm_batch(frames, D, 0.3, 6, 64, paths=8)
    for s in (0, 7):
        bd, mc = oracle.calc_cost_sgm(*frames[s][:2], D, 0.3, *frames[s][2:], 6, 64, 8)
        np.testing.assert_array_equal(res[s][1], mc, err_msg=f"frame {s} minC vs oracle")
        np.testing.assert_array_equal(res[s][0], bd, err_msg=f"frame {s} bestD vs oracle")
    for s in range(B):
        bd, mc = calc_cost_sgm(*frames[s][:2], D, 0.3, *frames[s][2:], 6, 64, paths=8)
        np.testing.assert_array_equal(res[s][1], mc, err_msg=f"frame {s} minC vs single call")
        np.testing.assert_array_equal(res[s][0], bd, err_msg=f"frame {s} bestD vs single call")
    with EpiPlan(W, H, D, B, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        for s in range(B):
            plan.upload(s, *frames[s])
        plan.set_agg_mode(2)
        plan.run()
        assert plan.kernel_name == "sweep16/nowrap"
        for s in range(B):
            gbd, gmc = plan.download(s)
            np.testing.assert_array_equal(gmc, res[s][1], err_msg=f"frame {s} minC, fused sweeps")
            np.testing.assert_array_equal(gbd, res[s][0], err_msg=f"frame {s} bestD, fused sweeps")


def test_cost_bound_reset_before_a_fused_run(gpu_lib, oracle):
    """upload_cost of a 255-valued volume selects the wrapping line kernels; a following upload of images and a run of
    all stages rewrites C with census costs (<= 24) and must come out of the fused sweeps, buffers in place."""
    W, H, D, B = 96, 40, 64, 8
    I1, I2 = synth.image_pair(W, H, D, seed=9)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=9)
    bd, mc = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, 8)
    with EpiPlan(W, H, D, B, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.upload_cost(0, np.full((H, W, D), 255, np.uint8))
        assert plan.kernel_name == "packed16/wrap"
        for f in range(B):
            plan.upload(f, I1, I2, pd0, nd, off)
        plan.set_agg_mode(2)
        plan.run()
        assert plan.kernel_name == "sweep16/nowrap"
        for f in (0, B - 1):
            gbd, gmc = plan.download(f)
            np.testing.assert_array_equal(gmc, mc)
            np.testing.assert_array_equal(gbd, bd)


# ------------------------------------------------------------------------------------------- config 4, full-window matcher
@pytest.mark.parametrize("W,H", [(1242, 375), (621, 188)])
def test_full_window_pyd_at_the_pyramid_level_sizes(gpu_lib, oracle, monkeypatch, W, H):
    """SURVEY 8(d) reads config 4 as calc_pyd_cost_sgm at 1242x375 / 621x188 / 311x94 with pyramidal_sgm.m:50's arguments
    (11x11 window, aggregation radius 2, 8 paths, 2 passes, P1=6, P2=32).  The two upper sizes against the oracle
    (calc_pyd_cost_sgm.cpp:114-372, :374-437), all voxels of C and S, bestD, minC, mvSub; integer and fractional hints;
    the aggregation kernel's mapping as the library picks it and both forced forms (the wide-rows mapping is chosen by
    frame shape, 1242-step lines, padded 132-byte pixels).  ~25 s of oracle per 1242x375 call."""
    I1, I2 = synth.image_pair(W, H, 16, seed=W)
    for kind, sub in (("int", 1), ("general", 1)):
        mv = synth.hint_map(W + 1, H + 2, kind, seed=H + len(kind))
        bd, mc, ms, Cv, S = oracle.calc_pyd_cost_sgm(I1, I2, mv, 5, 5, 2, sub, 6, 32, 1, 2, 0, want_volumes=True)
        for wide in (None, "0", "2"):
            if wide is None:
                monkeypatch.delenv("FSGM_PYD_WIDE", raising=False)
            else:
                monkeypatch.setenv("FSGM_PYD_WIDE", wide)
            gbd, gmc, gms, gC, gS = calc_pyd_cost_sgm(I1, I2, mv, 5, 5, 2, sub, 6, 32, 1, 2, 0, return_volumes=True)
            tag = f"hints {kind}, FSGM_PYD_WIDE={wide}"
            np.testing.assert_array_equal(gC, Cv, err_msg=tag)
            np.testing.assert_array_equal(gS, S, err_msg=tag)
            np.testing.assert_array_equal(gbd, bd, err_msg=tag)
            np.testing.assert_array_equal(gmc, mc, err_msg=tag)
            np.testing.assert_array_equal(gms, ms, err_msg=tag)
            del gC, gS
        del Cv, S


# ------------------------------------------------------------------------------------------- config 3 as shipped (4 paths)
def test_calc_cost_sgm_kitti_shape_4_paths_as_shipped(gpu_lib, oracle):
    """The configuration epipolar_sgm_of.m:45 actually runs: enableDiagnalPath = false (calc_cost_sgm.cpp:104), at
    1242x375x128.  One host call (the line kernels) with C and S, and a batch of 9 resident pairs (the pair pipeline)
    -- C, S, bestD, minC of every voxel / pixel against the oracle."""
    W, H, D = 1242, 375, 128
    I1, I2 = synth.image_pair(W, H, D, seed=77)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=78)
    bd, mc, Cv, S = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, 4, want_volumes=True)
    gbd, gmc, gC, gS = calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths=4, return_volumes=True)
    np.testing.assert_array_equal(gC, Cv)
    np.testing.assert_array_equal(gS, S)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gbd, bd)
    del gC, gS
    B = 9
    I1b, I2b = synth.image_pair(W, H, D, seed=79)
    pd0b, ndb, offb = synth.epi_maps(W, H, "axis", seed=80)
    bdb, mcb = oracle.calc_cost_sgm(I1b, I2b, D, 0.3, pd0b, ndb, offb, 6, 64, 4)
    with EpiPlan(W, H, D, B, paths=4) as plan:
        plan.set_penalties(6, 64, 0.3)
        for f in range(B):
            if f in (0, B - 1):
                plan.upload(f, I1, I2, pd0, nd, off)
            else:
                plan.upload(f, I1b, I2b, pd0b, ndb, offb)
        plan.run()
        assert plan.kernel_name == "pairs16/nowrap"
        for f in range(B):
            pbd, pmc = plan.download(f)
            wbd, wmc = (bd, mc) if f in (0, B - 1) else (bdb, mcb)
            np.testing.assert_array_equal(pmc, wmc, err_msg=f"frame {f} minC, pair pipeline")
            np.testing.assert_array_equal(pbd, wbd, err_msg=f"frame {f} bestD, pair pipeline")
        np.testing.assert_array_equal(plan.download_sum(0), S, err_msg="S rebuilt from the pair pipeline's volumes")
