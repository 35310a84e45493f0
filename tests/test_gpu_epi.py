"""GPU parity tests for the calc_cost_sgm path: HIP kernels (through the C ABI) vs the CPU oracle
on the same seeded inputs.  Integer / index outputs must be bit-exact.

Reference semantics under test: calc_cost_sgm.cpp:33-66 (sgm_step), :86-316 (sgm),
:319-412 (calc_cost), :414-426 (convert_vzInd_to_disp), common.cpp:3-27 (census).
"""
import numpy as np
import pytest

from fsgm_amd import synth, EpiPlan, calc_cost_sgm, calc_cost_sgm_batch
from fsgm_amd._lib import STAGE_COST, STAGE_AGGREGATE, STAGE_WTA

pytestmark = pytest.mark.gpu


def _last_pixel_mask(H, W):
    m = np.ones((H, W), bool)
    return m


# ---------------------------------------------------------------- cost volume (census+Hamming+box)
@pytest.mark.parametrize("W,H,D,kind", [
    (64, 48, 16, "general"), (67, 45, 32, "general"), (40, 30, 64, "axis"),
    (33, 21, 128, "general"), (31, 17, 20, "general"), (29, 19, 7, "general"),
    (35, 27, 136, "general"), (23, 9, 8, "general"), (50, 21, 264, "axis"), (128, 3, 24, "general"), (97, 55, 64, "radial"),   # pixel-per-thread kernel: D segments, ragged last wave
])
def test_cost_volume_bit_exact(gpu_lib, oracle, W, H, D, kind):
    I1, I2 = synth.image_pair(W, H, D, seed=W + D)
    pd0, nd, off = synth.epi_maps(W, H, kind, seed=H)
    want = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    with EpiPlan(W, H, D, 1, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.upload(0, I1, I2, pd0, nd, off)
        plan.run(STAGE_COST)
        got = plan.download_cost(0)
    assert got.max() <= 24
    np.testing.assert_array_equal(got, want)


def test_cost_volume_out_of_range_geometry(gpu_lib, oracle):
    """Sample positions far outside the image / non-finite: the reference's (int) conversion gives
    INT_MIN on x86-64, which clamps to 0 -- not the saturating GPU conversion."""
    W, H, D = 48, 32, 16
    I1, I2 = synth.image_pair(W, H, D, seed=5)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=9)
    off[3, 4] = 1e12
    off[5, 6] = -1e12
    off[7, 8] = 1e300
    pd0[0, 9, 9] = np.inf
    pd0[1, 10, 10] = np.nan
    want = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    with EpiPlan(W, H, D, 1) as plan:
        plan.upload(0, I1, I2, pd0, nd, off)
        plan.run(STAGE_COST)
        got = plan.download_cost(0)
    np.testing.assert_array_equal(got, want)


# ---------------------------------------------------------------- aggregation
AGG_CASES = [
    # W, H, D, P1, P2, cmax, expected kernel
    (64, 48, 16, 6, 64, 24, "packed16/nowrap"),
    (67, 45, 32, 6, 32, 24, "packed16/nowrap"),
    (41, 29, 64, 6, 64, 24, "packed16/nowrap"),
    (37, 23, 128, 6, 64, 24, "packed16/nowrap"),
    (19, 11, 256, 3, 20, 24, "packed16/nowrap"),
    (64, 48, 16, 100, 200, 255, "packed16/wrap"),
    (37, 23, 128, 90, 120, 255, "packed16/wrap"),
    (45, 31, 64, 6, 64, 255, "packed16/wrap"),
    (30, 17, 32, 70, 90, 255, "packed16/wrap"),
    (23, 13, 256, 90, 120, 255, "packed16/wrap"),
    (131, 9, 128, 6, 64, 24, "packed16/nowrap"),             # along-x lines past two workgroups of the fine split
    (33, 21, 20, 6, 64, 24, "generic"),
    (21, 17, 7, 100, 200, 255, "generic"),
    (9, 5, 1, 6, 64, 24, "generic"),
]


@pytest.mark.parametrize("paths", [4, 8])
@pytest.mark.parametrize("W,H,D,P1,P2,cmax,kernel", AGG_CASES)
def test_aggregate_sum_bit_exact(gpu_lib, oracle, W, H, D, P1, P2, cmax, kernel, paths):
    Cv = synth.cost_volume(W, H, D, seed=W * 7 + D, cmax=cmax)
    want = oracle.epi_aggregate(Cv, P1, P2, paths)[:-1].reshape(H, W, D)
    with EpiPlan(W, H, D, 1, paths=paths) as plan:
        plan.set_penalties(P1, P2, 0.3)
        plan.upload_cost(0, Cv)
        # mode 2: the fused sweeps take the 8-path no-wrap case, the pair kernels the 4-path one (the shipped
        # configuration), the line kernels everything else
        assert plan.kernel_name == kernel                   # auto mode, 1 frame: line kernels
        plan.set_agg_mode(2)
        fused = {8: "sweep16/nowrap", 4: "pairs16/nowrap"}[paths]
        assert plan.kernel_name == (fused if kernel == "packed16/nowrap" else kernel)
        plan.run(STAGE_AGGREGATE)
        got = plan.download_sum(0)
        np.testing.assert_array_equal(got, want)
        plan.set_agg_mode(1)                                # per-direction line kernels
        assert plan.kernel_name == kernel
        plan.run(STAGE_AGGREGATE)
        got = plan.download_sum(0)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("W,H,D,B", [(70, 40, 128, 2), (200, 53, 64, 1), (33, 100, 128, 3), (50, 20, 256, 1), (257, 19, 16, 2), (40, 300, 32, 1),
                                     # sizes around the 8-position tiles of either axis (no / one / partial checkpoint)
                                     (1, 9, 64, 1), (9, 1, 64, 1), (5, 20, 128, 2), (8, 8, 128, 1), (9, 17, 32, 1), (16, 7, 128, 1),
                                     (17, 16, 64, 2), (1242, 9, 128, 1), (12, 375, 128, 1),
                                     # 16 frames and more: the along-x pair with 16 costs a lane (8 a lane below)
                                     (33, 21, 64, 16), (40, 12, 128, 17)])
def test_pairs_pipeline_4_paths(gpu_lib, oracle, W, H, D, B):
    """The 4-path pair pipeline (horizontal pair -> X_h, vertical pair final with the WTA): S through the
    debug tap, bestD / minC through the records, against the oracle and the line kernels."""
    vols = [synth.cost_volume(W, H, D, seed=W + H + f, cmax=24) for f in range(B)]
    for v in vols:
        v[:, ::5, :] = 0
    _, _, off = synth.epi_maps(W, H, "general", seed=3)
    with EpiPlan(W, H, D, B, paths=4) as plan:
        plan.set_penalties(6, 64, 0.3)
        for f in range(B):
            plan.upload_cost(f, vols[f])
            plan.upload_offset(f, off)
        plan.set_agg_mode(2)
        assert plan.kernel_name == "pairs16/nowrap"
        for _ in range(2):                                   # twice: scratch buffers are reused
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
        for f in range(B):
            S = oracle.epi_aggregate(vols[f], 6, 64, 4)
            bd, mc = oracle.epi_wta(S, W, H, D, 1)
            gbd, gmc = plan.download(f)
            np.testing.assert_array_equal(gmc, mc)
            np.testing.assert_array_equal(gbd, oracle.epi_vz_to_disp(bd, off, 0.3, D + 1))
            np.testing.assert_array_equal(plan.download_sum(f), S[:-1].reshape(H, W, D))


@pytest.mark.parametrize("W,H,D", [(70, 40, 128), (200, 53, 64), (33, 100, 128), (130, 35, 32), (50, 20, 256), (257, 19, 16),
                                   (320, 240, 64), (40, 300, 16), (60, 150, 32), (23, 40, 256),
                                   # widths around the horizontal pair's 8-column tiles (no / one / partial checkpoint)
                                   (1, 9, 64), (5, 20, 128), (8, 33, 128), (9, 20, 32), (16, 20, 128), (17, 40, 64), (1242, 17, 128)])
def test_sweep_blocks_and_columns(gpu_lib, oracle, W, H, D):
    """Fused sweeps across several row blocks and column strips of workgroups (block / halo / state hand-over paths)."""
    Cv = synth.cost_volume(W, H, D, seed=W + H, cmax=24)
    Cv[:, ::7, :] = 0                                        # strong structure so diagonals carry information far
    want = oracle.epi_aggregate(Cv, 6, 64, 8)[:-1].reshape(H, W, D)
    with EpiPlan(W, H, D, 2, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.upload_cost(0, Cv)
        plan.upload_cost(1, np.ascontiguousarray(Cv[::-1]))
        plan.set_agg_mode(2)
        assert plan.kernel_name == "sweep16/nowrap"
        plan.run(STAGE_AGGREGATE)
        got = plan.download_sum(0)
        got1 = plan.download_sum(1)
    np.testing.assert_array_equal(got, want)
    want1 = oracle.epi_aggregate(np.ascontiguousarray(Cv[::-1]), 6, 64, 8)[:-1].reshape(H, W, D)
    np.testing.assert_array_equal(got1, want1)


def test_aggregate_tiny_and_degenerate_shapes(gpu_lib, oracle):
    """1-pixel-wide / 1-pixel-high images: every diagonal step is a path start."""
    for (W, H) in [(1, 1), (1, 9), (9, 1), (2, 2), (8, 3), (3, 8)]:
        for D in (16, 128):
            Cv = synth.cost_volume(W, H, D, seed=W + 10 * H, cmax=24)
            want = oracle.epi_aggregate(Cv, 6, 64, 8)[:-1].reshape(H, W, D)
            with EpiPlan(W, H, D, 1, paths=8) as plan:
                plan.upload_cost(0, Cv)
                plan.run(STAGE_AGGREGATE)
                got = plan.download_sum(0)
            np.testing.assert_array_equal(got, want, err_msg=f"{W}x{H}x{D}")


# ---------------------------------------------------------------- WTA / sub-pixel / vz
@pytest.mark.parametrize("W,H,D", [(64, 48, 16), (37, 23, 128), (33, 21, 20)])
@pytest.mark.parametrize("subpixel,vz", [(1, 1), (1, 0), (0, 0)])
def test_wta_subpixel_bit_exact(gpu_lib, oracle, W, H, D, subpixel, vz):
    Cv = synth.cost_volume(W, H, D, seed=11, cmax=24)
    # force the best == D-1 (reads next pixel's d=0), best == 1 (never refined) and best == 0 cases
    Cv[2, 3, :] = 24; Cv[2, 3, D - 1] = 0
    Cv[4, 5, :] = 24; Cv[4, 5, 1] = 0
    Cv[6, 7, :] = 24; Cv[6, 7, 0] = 0
    Cv[H - 1, W - 1, :] = 24; Cv[H - 1, W - 1, D - 1] = 0      # last pixel: Sp[best+1] is past the array -> 0
    _, _, off = synth.epi_maps(W, H, "general", seed=3)
    S = oracle.epi_aggregate(Cv, 6, 64, 8)
    bd, mc = oracle.epi_wta(S, W, H, D, subpixel)
    if vz:
        bd = oracle.epi_vz_to_disp(bd, off, 0.3, D + 1)
    with EpiPlan(W, H, D, 1, paths=8, subpixel=subpixel, vz_to_disp=vz) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.upload_cost(0, Cv)
        plan.upload_offset(0, off)
        for mode in (1, 2):                                  # line kernels + WTA kernel / sweeps with fused WTA
            plan.set_agg_mode(mode)
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
            gbd, gmc = plan.download(0)
            np.testing.assert_array_equal(gmc, mc, err_msg=plan.kernel_name)
            np.testing.assert_array_equal(gbd, bd, err_msg=plan.kernel_name)


# ---------------------------------------------------------------- whole MEX
@pytest.mark.parametrize("paths", [4, 8])
@pytest.mark.parametrize("W,H,D,kind", [(64, 48, 16, "general"), (320, 240, 64, "axis"), (97, 61, 128, "general"),
                                        (50, 40, 24, "general")])
def test_calc_cost_sgm_whole_mex(gpu_lib, oracle, W, H, D, kind, paths):
    I1, I2 = synth.image_pair(W, H, D, seed=2)
    pd0, nd, off = synth.epi_maps(W, H, kind, seed=4)
    bd, mc, Cv, S = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths, want_volumes=True)
    gbd, gmc, gC, gS = calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths=paths, return_volumes=True)
    np.testing.assert_array_equal(gC, Cv)
    np.testing.assert_array_equal(gS, S)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gbd, bd)


def test_calc_cost_sgm_kitti_shape_8_paths(gpu_lib, oracle):
    """BASELINE config 3: 1242x375, D=128, 8 paths -- full-size bit-exact comparison, through the
    host entry point (one frame: line kernels) and through a plan forced onto the fused sweeps."""
    W, H, D = 1242, 375, 128
    I1, I2 = synth.image_pair(W, H, D, seed=1)
    pd0, nd, off = synth.epi_maps(W, H, "axis")
    bd, mc, Cv, S = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, 8, want_volumes=True)
    gbd, gmc, gC, gS = calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths=8, return_volumes=True)
    np.testing.assert_array_equal(gC, Cv)
    np.testing.assert_array_equal(gS, S)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gbd, bd)
    with EpiPlan(W, H, D, 1, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.set_agg_mode(2)
        plan.upload(0, I1, I2, pd0, nd, off)
        plan.run()
        assert plan.kernel_name == "sweep16/nowrap"
        sbd, smc = plan.download(0)
        np.testing.assert_array_equal(smc, mc)
        np.testing.assert_array_equal(sbd, bd)
        np.testing.assert_array_equal(plan.download_sum(0), S)


def test_cost_volume_kitti_shape_general_field(gpu_lib, oracle):
    """Full-size cost volume on the general direction field (fractional positions, exact .5 ties, a random
    direction per pixel): the pixel-per-thread cost fill against the oracle, all 59.6 M voxels."""
    W, H, D = 1242, 375, 128
    I1, I2 = synth.image_pair(W, H, D, seed=2)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=4)
    want = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    with EpiPlan(W, H, D, 1, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.upload(0, I1, I2, pd0, nd, off)
        plan.run(STAGE_COST)
        np.testing.assert_array_equal(plan.download_cost(0), want)


# (auto mode: frames this small stay on the line kernels at every batch size -- the fused pipelines through the batch host call
#  are covered at the KITTI shape, tests/test_gpu_configs_full_size.py)
@pytest.mark.parametrize("paths,n", [(8, 18), (4, 9), (8, 5), (8, 3)])
def test_batch_matches_single_frames(gpu_lib, oracle, paths, n):
    W, H, D = 96, 64, 64
    frames = []
    for s in range(n):
        I1, I2 = synth.image_pair(W, H, D, seed=20 + s)
        pd0, nd, off = synth.epi_maps(W, H, "general", seed=30 + s)
        frames.append((I1, I2, pd0, nd, off))
    res = calc_cost_sgm_batch(frames, D, 0.3, 6, 64, paths=paths)
    for (I1, I2, pd0, nd, off), (gbd, gmc) in zip(frames, res):
        bd, mc = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths)
        np.testing.assert_array_equal(gmc, mc)
        np.testing.assert_array_equal(gbd, bd)


@pytest.mark.parametrize("paths,kernel", [(8, "sweep16/nowrap"), (4, "pairs16/nowrap")])
def test_full_size_property_mirror_symmetry(gpu_lib, paths, kernel):
    """Size-independent properties at full KITTI size, for both fused pipelines: point-mirroring the cost
    volume in (x,y) mirrors S (pass 1 of the reference is the point mirror of pass 0,
    calc_cost_sgm.cpp:115-123); paths*C <= S <= paths*(C + P2) elementwise when nothing wraps; and the
    fused pipeline agrees with the per-direction line kernels voxel for voxel."""
    W, H, D = 1242, 375, 128
    Cv = synth.cost_volume(W, H, D, seed=99, cmax=24)
    with EpiPlan(W, H, D, 2, paths=paths) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.set_agg_mode(2)
        assert plan.kernel_name == kernel
        plan.upload_cost(0, Cv)
        plan.upload_cost(1, np.ascontiguousarray(Cv[::-1, ::-1, :]))
        plan.run(STAGE_AGGREGATE)
        S0 = plan.download_sum(0)
        S1 = plan.download_sum(1)
        plan.set_agg_mode(1)
        plan.run(STAGE_AGGREGATE)
        L0 = plan.download_sum(0)
    np.testing.assert_array_equal(S1[::-1, ::-1, :], S0)
    np.testing.assert_array_equal(L0, S0)
    assert (S0 >= paths * Cv.astype(np.uint32)).all()           # every L_r(p,d) >= C(p,d) when nothing wraps
    assert (S0 <= paths * (Cv.astype(np.uint32) + 64)).all()    # and <= C + P2


def test_run_sharded_with_the_hip_compute(gpu_lib, oracle):
    """fsgm_amd.batch.run_sharded on one rank with the real (HIP) per-shard compute."""
    from fsgm_amd import batch
    W, H, D = 64, 40, 32
    frames = []
    for s in range(3):
        I1, I2 = synth.image_pair(W, H, D, seed=70 + s)
        pd0, nd, off = synth.epi_maps(W, H, "general", seed=80 + s)
        frames.append((I1, I2, pd0, nd, off))
    res = batch.run_sharded(frames, lambda fs: calc_cost_sgm_batch(fs, D, 0.3, 6, 64, paths=8, device=0), rank=0, world=1)
    for (I1, I2, pd0, nd, off), (gbd, gmc) in zip(frames, res):
        bd, mc = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, 8)
        np.testing.assert_array_equal(gbd, bd)
        np.testing.assert_array_equal(gmc, mc)


def test_cost_volume_rounding_edge_cases(gpu_lib, oracle):
    """Sample positions engineered onto the hard cases of (int)round(v): exact halves of both signs,
    0.5 - 1ulp (where floor(v+0.5) would be wrong), values just inside / at / beyond +-2^31."""
    W, H, D = 40, 24, 16
    I1, I2 = synth.image_pair(W, H, D, seed=8)
    pd0, nd, off = synth.epi_maps(W, H, "axis")
    nd[0][:] = 1.0; nd[1][:] = 0.0
    off[:] = 0.0                                             # sample position = Pd0 - 1 exactly
    specials = [0.5, 1.5, 2.5, -0.5, -1.5, 0.49999999999999994, -0.49999999999999994, 1.4999999999999998,
                2147483647.5, 2147483647.4, 2147483648.0, -2147483648.0, -2147483648.5, -2147483649.0,
                4294967296.0, 1e300, -1e300, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 38.5, 39.5, 39.49999999999999]
    for i, v in enumerate(specials):
        pd0[0, i % H, (3 * i) % W] = v + 1.0 if np.isfinite(v) and abs(v) < 1e15 else v
        pd0[1, (i + 5) % H, (3 * i + 1) % W] = v + 1.0 if np.isfinite(v) and abs(v) < 1e15 else v
    want = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    with EpiPlan(W, H, D, 1) as plan:
        plan.upload(0, I1, I2, pd0, nd, off)
        plan.run(STAGE_COST)
        got = plan.download_cost(0)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("W,H,D,kind,paths", [(64, 48, 16, "axis", 8), (97, 61, 64, "general", 8), (50, 40, 24, "general", 4),
                                               (160, 90, 128, "axis", 8)])
def test_forward_backward_check(gpu_lib, oracle, W, H, D, kind, paths):
    """fb_check=1: forward_backward_check / calc_disp_from_first (calc_cost_sgm.cpp:429-536, dead code
    in the shipped reference) on bestD before vz->disparity; conf and bestD2 bit-exact, bestD/minC unchanged."""
    I1, I2 = synth.image_pair(W, H, D, seed=5)
    pd0, nd, off = synth.epi_maps(W, H, kind, seed=6)
    Cv = oracle.epi_cost(I1, I2, D, 0.3, pd0, nd, off)
    S = oracle.epi_aggregate(Cv, 6, 64, paths)
    bd_idx, mc = oracle.epi_wta(S, W, H, D, 1)
    conf, d2 = oracle.epi_fb_check(bd_idx, pd0, nd, off, 0.3, D + 1)
    bd = oracle.epi_vz_to_disp(bd_idx, off, 0.3, D + 1)
    gbd, gmc, gconf, gd2 = calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths=paths, fb_check=1)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gbd, bd)
    np.testing.assert_array_equal(gd2, d2)
    np.testing.assert_array_equal(gconf, conf)
    assert 0 < int(gconf.sum()) < W * H                      # both outcomes occur
    # batches of 5 and 18 go through the parallel sweeps and the full sweep pipeline (8 paths; the line and the pair kernels
    # for 4), with the volumes read back through the debug taps: same answer
    for n in (5, 18):
        res = calc_cost_sgm_batch([(I1, I2, pd0, nd, off)] * n, D, 0.3, 6, 64, paths=paths, fb_check=1, return_volumes=(n == 5))
        for r in res:
            np.testing.assert_array_equal(r[0], bd)
            np.testing.assert_array_equal(r[-2], conf)
            np.testing.assert_array_equal(r[-1], d2)
            if n == 5:
                np.testing.assert_array_equal(r[2], Cv)
                np.testing.assert_array_equal(r[3], S[:-1].reshape(H, W, D))


def test_sweep_pipeline_is_deterministic_under_back_to_back_runs(gpu_lib, oracle):
    """The fused-sweep stage runs on three streams with event fork/join; back-to-back runs reuse
    every intermediate buffer.  30 runs without host synchronisation in between must leave exactly
    the oracle's answer in every frame (a missing dependency would show as a rare wrong frame)."""
    W, H, D, B = 150, 70, 128, 6
    vols = [synth.cost_volume(W, H, D, seed=40 + f, cmax=24) for f in range(B)]
    _, _, off = synth.epi_maps(W, H, "general", seed=1)
    want = []
    for v in vols:
        S = oracle.epi_aggregate(v, 6, 64, 8)
        bd, mc = oracle.epi_wta(S, W, H, D, 1)
        want.append((oracle.epi_vz_to_disp(bd, off, 0.3, D + 1), mc))
    with EpiPlan(W, H, D, B, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.set_agg_mode(2)
        assert plan.kernel_name == "sweep16/nowrap"
        for f in range(B):
            plan.upload_cost(f, vols[f])
            plan.upload_offset(f, off)
        for rep in range(3):
            for _ in range(10):
                plan.run(STAGE_AGGREGATE | STAGE_WTA)
            for f in range(B):
                gbd, gmc = plan.download(f)
                np.testing.assert_array_equal(gmc, want[f][1], err_msg=f"rep {rep} frame {f}")
                np.testing.assert_array_equal(gbd, want[f][0], err_msg=f"rep {rep} frame {f}")


def test_one_plan_through_every_pipeline(gpu_lib, oracle):
    """The pipelines share buffers (records, Y volumes, the pair's stream) that are created on first use: one plan switched
    through every aggregation mode in both orders must give the oracle's result each time (8 and 4 paths)."""
    W, H, D, B = 96, 70, 128, 3
    vols = [synth.cost_volume(W, H, D, seed=31 + f, cmax=24) for f in range(B)]
    _, _, off = synth.epi_maps(W, H, "general", seed=3)
    for paths, modes in ((8, (4, 2, 3, 6, 5, 1, 6, 5, 3, 2, 6, 4)), (4, (4, 2, 5, 1, 2, 4))):
        want = []
        for v in vols:
            S = oracle.epi_aggregate(v, 6, 64, paths)
            bd, mc = oracle.epi_wta(S, W, H, D, 1)
            want.append((oracle.epi_vz_to_disp(bd, off, 0.3, D + 1), mc, S[:-1].reshape(H, W, D)))
        with EpiPlan(W, H, D, B, paths=paths) as plan:
            plan.set_penalties(6, 64, 0.3)
            for f in range(B):
                plan.upload_cost(f, vols[f])
                plan.upload_offset(f, off)
            for mode in modes:
                plan.set_agg_mode(mode)
                plan.run(STAGE_AGGREGATE | STAGE_WTA)
                for f in range(B):
                    gbd, gmc = plan.download(f)
                    np.testing.assert_array_equal(gmc, want[f][1], err_msg=f"{paths} paths, mode {mode} ({plan.kernel_name}), frame {f}")
                    np.testing.assert_array_equal(gbd, want[f][0], err_msg=f"{paths} paths, mode {mode} ({plan.kernel_name}), frame {f}")
                np.testing.assert_array_equal(plan.download_sum(B - 1), want[B - 1][2], err_msg=f"{paths} paths, mode {mode}: S")


@pytest.mark.parametrize("subpixel", [1, 0])
@pytest.mark.parametrize("W,H,D,B", [(150, 70, 128, 3), (97, 200, 64, 2), (1242, 40, 128, 1), (33, 375, 128, 2), (20, 9, 16, 1), (260, 31, 32, 2),
                                     (70, 50, 256, 1), (1, 9, 64, 1), (9, 1, 64, 2), (96, 64, 64, 7), (45, 33, 128, 2), (40, 2, 128, 1), (64, 97, 32, 1)])
def test_parallel_sweeps_match_the_oracle(gpu_lib, oracle, W, H, D, B, subpixel):
    """Aggregation mode 3 (down and up sweeps side by side, Y_up written out, WTA over C, Y_dn, Y_up, Y_h) and mode 6 (the two
    sweeps meet in the middle and finish each other's half with the WTA inside): bestD / minC of every frame and S of the last
    frame against the oracle, back-to-back runs on reused buffers; then the same plan in mode 2 (buffers shared between the forms)."""
    vols = [synth.cost_volume(W, H, D, seed=W + H + f, cmax=24) for f in range(B)]
    _, _, off = synth.epi_maps(W, H, "general", seed=3)
    want, S = [], None
    for v in vols:
        S = oracle.epi_aggregate(v, 6, 64, 8)
        bd, mc = oracle.epi_wta(S, W, H, D, subpixel)
        want.append((oracle.epi_vz_to_disp(bd, off, 0.3, D + 1), mc))
    with EpiPlan(W, H, D, B, paths=8, subpixel=subpixel) as plan:
        plan.set_penalties(6, 64, 0.3)
        for f in range(B):
            plan.upload_cost(f, vols[f])
            plan.upload_offset(f, off)
        for mode, name in ((3, "sweep16par/nowrap"), (6, "sweep16mid/nowrap"), (2, "sweep16/nowrap"), (6, "sweep16mid/nowrap"), (3, "sweep16par/nowrap")):
            plan.set_agg_mode(mode)
            assert plan.kernel_name == name
            for _ in range(3):
                plan.run(STAGE_AGGREGATE | STAGE_WTA)
            for f in range(B):
                gbd, gmc = plan.download(f)
                np.testing.assert_array_equal(gmc, want[f][1], err_msg=f"mode {mode} frame {f}")
                np.testing.assert_array_equal(gbd, want[f][0], err_msg=f"mode {mode} frame {f}")
            np.testing.assert_array_equal(plan.download_sum(B - 1), S[:-1].reshape(H, W, D), err_msg=f"mode {mode}")


@pytest.mark.parametrize("paths", [8, 4])
@pytest.mark.parametrize("W,H,D", [(1, 1, 128), (2, 5, 128), (3, 2, 128), (31, 3, 128), (32, 4, 128), (33, 9, 128), (34, 2, 128), (35, 40, 128),
                                   (63, 5, 128), (64, 6, 128), (65, 7, 128), (66, 3, 128), (67, 4, 128), (97, 11, 128), (130, 9, 128), (200, 3, 128),
                                   (7, 40, 32), (40, 7, 32), (5, 33, 64), (9, 70, 256), (70, 9, 256), (3, 100, 128), (1, 50, 64),
                                   (66, 3, 64), (97, 5, 64), (200, 7, 64), (33, 2, 64), (130, 6, 32), (65, 9, 32), (98, 5, 32), (34, 1, 32), (320, 11, 64)])
def test_line_kernels_hand_written_steps_at_their_loop_boundaries(gpu_lib, oracle, W, H, D, paths):
    """The line kernels' hand-written steps (agg_x_lean_body: 32 steps of C in flight, a main loop and two tails; agg_lean_body: 4 in
    flight, the diagonals' wrap by a down-counter): line lengths around every loop bound, lines shorter than the prefetch depth,
    frames narrower than they are tall (a diagonal wraps several times), two frames."""
    vols = [synth.cost_volume(W, H, D, seed=3 * W + H + f, cmax=24) for f in range(2)]
    with EpiPlan(W, H, D, 2, paths=paths, subpixel=1, vz_to_disp=0) as plan:
        plan.set_penalties(6, 64, 0.3)
        for f in range(2):
            plan.upload_cost(f, vols[f])
        plan.set_agg_mode(1)
        assert plan.kernel_name == "packed16/nowrap"
        plan.run(STAGE_AGGREGATE | STAGE_WTA)
        for f in range(2):
            S = oracle.epi_aggregate(vols[f], 6, 64, paths)
            bd, mc = oracle.epi_wta(S, W, H, D, 1)
            gbd, gmc = plan.download(f)
            np.testing.assert_array_equal(plan.download_sum(f), S[:-1].reshape(H, W, D))
            np.testing.assert_array_equal(gmc, mc)
            np.testing.assert_array_equal(gbd, bd)


def test_auto_mode_by_batch_size(gpu_lib):
    """A plan in auto mode runs what fsgm_epi_auto_pipeline says (the table itself: tests/test_capi_cpu.py); small frames stay
    on the line kernels far beyond the KITTI shape's switch points; wrapping penalties always take the line kernels."""
    from fsgm_amd import auto_pipeline
    for (W, H, D), paths, B in [((32, 16, 64), 8, 3), ((32, 16, 64), 8, 40), ((32, 16, 64), 4, 9), ((320, 240, 64), 8, 24), ((320, 240, 64), 4, 48)]:
        with EpiPlan(W, H, D, B, paths=paths) as plan:
            plan.set_penalties(6, 64, 0.3)
            assert plan.kernel_name == auto_pipeline(W, H, D, B, paths, 6, 64), (W, H, D, paths, B)
            if W == 32:
                assert plan.kernel_name == "packed16/nowrap"
            plan.set_penalties(100, 200, 0.3)
            assert plan.kernel_name == "packed16/wrap"


def _sharded_gpu_worker(rank, world, port, n_frames, q):
    """One rank of test_two_processes_shard_frames_on_the_gpu: started before anything touches the GPU in this process."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from fsgm_amd import batch, synth as sy, calc_cost_sgm_batch as run
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = []
        for s in range(n_frames):
            I1, I2 = sy.image_pair(96, 64, 64, seed=70 + s)
            frames.append((I1, I2) + sy.epi_maps(96, 64, "general", seed=80 + s))
        full = batch.run_sharded(frames, lambda fs: run(fs, 64, 0.3, 6, 64, paths=8, device=0))   # both ranks on the one GPU of the box
        q.put((rank, [(bd.copy(), mc.copy()) for bd, mc in full]))
    finally:
        dist.destroy_process_group()


def test_two_processes_shard_frames_on_the_gpu(gpu_lib, oracle):
    """The N > 1 path with the real compute: two worker processes (gloo rendezvous, both on device 0 -- the box has one
    GPU) run fsgm_amd.batch.run_sharded over calc_cost_sgm_batch; every rank must end up with the oracle's result for
    every frame, in frame order."""
    import socket
    import torch.multiprocessing as mp
    n_frames, world = 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_gpu_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for s in range(n_frames):
        I1, I2 = synth.image_pair(96, 64, 64, seed=70 + s)
        pd0, nd, off = synth.epi_maps(96, 64, "general", seed=80 + s)
        bd, mc = oracle.calc_cost_sgm(I1, I2, 64, 0.3, pd0, nd, off, 6, 64, 8)
        for r in range(world):
            np.testing.assert_array_equal(got[r][s][0], bd, err_msg=f"rank {r} frame {s}")
            np.testing.assert_array_equal(got[r][s][1], mc, err_msg=f"rank {r} frame {s}")


@pytest.mark.parametrize("W,H,D,P1,P2,paths", [(64, 48, 64, 6, 64, 4), (33, 21, 128, 6, 64, 8), (40, 30, 24, 7, 100, 4), (21, 17, 16, 100, 200, 8)])
def test_sgm_call_shape(gpu_lib, oracle, W, H, D, P1, P2, paths):
    """sgm(C, P1, P2) (sgm.m:1, test.m:36) served by the MEX's aggregation + WTA: S, minC and bestD (index * 256 with the
    MEX parabola, no vz conversion) against the oracle, incl. sgm.m's own defaults 7 / 100 and wrapping penalties."""
    from fsgm_amd import sgm
    Cv = synth.cost_volume(W, H, D, seed=W + D, cmax=24)
    want = oracle.epi_aggregate(Cv, P1, P2, paths)
    bd, mc = oracle.epi_wta(want, W, H, D, 1)
    gbd, gmc, gS = sgm(Cv, P1, P2, paths=paths, return_sum=True)
    np.testing.assert_array_equal(gS, want[:-1].reshape(H, W, D))
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gbd, bd)


def _coarse_split_worker(q):
    """The line kernels with 16 costs a lane along x too (FSGM_AGG_FINE=0, read once per process)."""
    import os
    os.environ["FSGM_AGG_FINE"] = "0"
    from fsgm_amd import sgm, synth as sy
    out = []
    for W, H, D, P1, P2, cmax in [(70, 21, 128, 6, 64, 24), (45, 31, 64, 90, 120, 255), (19, 11, 256, 3, 20, 24)]:
        Cv = sy.cost_volume(W, H, D, seed=W + D, cmax=cmax)
        out.append(sgm(Cv, P1, P2, paths=8, return_sum=True)[2].copy())
    q.put(out)


def test_line_kernels_coarse_split_matches_the_oracle(gpu_lib, oracle):
    """Both lane splits of agg_packed_kernel stay checked: the default (4 costs a lane along x) by every line-kernel
    test above, the 16-costs-a-lane split here, in a process of its own."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_coarse_split_worker, args=(q,))
    p.start()
    got = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    for (W, H, D, P1, P2, cmax), g in zip([(70, 21, 128, 6, 64, 24), (45, 31, 64, 90, 120, 255), (19, 11, 256, 3, 20, 24)], got):
        Cv = synth.cost_volume(W, H, D, seed=W + D, cmax=cmax)
        want = oracle.epi_aggregate(Cv, P1, P2, 8)[:-1].reshape(H, W, D)
        np.testing.assert_array_equal(g, want, err_msg=f"{W}x{H}x{D}")
