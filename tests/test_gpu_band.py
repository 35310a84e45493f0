"""GPU parity tests of the band sweeps (fsgm_amd/csrc/epi_band.hip): all four paths of a raster pass of
calc_cost_sgm.cpp:86-257 in one sweep over the skewed coordinate u = x + 2y (8 paths) / u = x + y (the shipped 4 paths,
:104), one workgroup per frame, band after band.  Against the CPU oracle: S of every voxel (the kernel's debug tap),
bestD and minC of every pixel; shapes chosen around the band height (64 rows at D = 128, 128 at D = 64, 512 at D = 16),
the image borders (path starts, :152-180), one-pixel-wide / -high frames, penalties with and without the 9th-bit plane
(4*P2 > 255)."""
import numpy as np
import pytest

from fsgm_amd import synth, EpiPlan
from fsgm_amd._lib import STAGE_AGGREGATE, STAGE_WTA

pytestmark = pytest.mark.gpu


def _check(oracle, plan, vols, off, P1, P2, paths, W, H, D, taps=(0,)):
    for f, v in enumerate(vols):
        S = oracle.epi_aggregate(v, P1, P2, paths)
        bd, mc = oracle.epi_wta(S, W, H, D, 1)
        gbd, gmc = plan.download(f)
        np.testing.assert_array_equal(gmc, mc, err_msg=f"frame {f} minC")
        np.testing.assert_array_equal(gbd, oracle.epi_vz_to_disp(bd, off, 0.3, D + 1), err_msg=f"frame {f} bestD")
        if f in taps:
            np.testing.assert_array_equal(plan.download_sum(f), S[:-1].reshape(H, W, D), err_msg=f"frame {f} S")


SHAPES = [
    # W, H, D, frames
    (40, 30, 128, 2),        # one band, partly filled
    (37, 64, 128, 1),        # exactly one band
    (29, 65, 128, 2),        # one row into the second band
    (130, 150, 128, 2),      # three bands, the last partial
    (70, 129, 64, 1),        # D = 64: bands of 128 rows
    (33, 260, 32, 1),        # D = 32: bands of 256 rows
    (21, 530, 16, 1),        # D = 16: bands of 512 rows
    (50, 20, 256, 2),        # D = 256: bands of 32 rows... still one band
    (45, 70, 256, 1),        # D = 256, three bands
    (1, 9, 128, 1), (9, 1, 128, 2), (1, 1, 64, 1), (2, 2, 128, 1), (3, 140, 128, 1), (140, 3, 64, 1),
    (320, 240, 64, 1),       # BASELINE configs[1] shape
]


@pytest.mark.parametrize("paths", [8, 4])
@pytest.mark.parametrize("W,H,D,B", SHAPES)
def test_band_sweeps_match_the_oracle(gpu_lib, oracle, W, H, D, B, paths):
    vols = [synth.cost_volume(W, H, D, seed=W * 3 + H + f, cmax=24) for f in range(B)]
    for v in vols:
        v[:, ::5, :] = 0                                     # strong structure: paths carry information far
    _, _, off = synth.epi_maps(W, H, "general", seed=3)
    with EpiPlan(W, H, D, B, paths=paths) as plan:
        plan.set_penalties(6, 64, 0.3)                       # 4 * (P1 + P2) > 255: the first pass's sums need their 9th bits
        for f in range(B):
            plan.upload_cost(f, vols[f])
            plan.upload_offset(f, off)
        plan.set_agg_mode(4)
        assert plan.kernel_name == "band16/nowrap"
        for _ in range(2):                                   # twice: the hand-off buffer and the volumes are reused
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
        _check(oracle, plan, vols, off, 6, 64, paths, W, H, D, taps=(0, B - 1))


@pytest.mark.parametrize("paths", [8, 4])
@pytest.mark.parametrize("P1,P2,cmax", [(6, 32, 24), (0, 0, 24), (10, 10, 60), (3, 63, 24), (6, 85, 24), (20, 100, 50), (1, 126, 0), (63, 64, 24)])
def test_band_sweeps_penalties(gpu_lib, oracle, P1, P2, cmax, paths):
    """With and without the bit plane (4*(P1+P2) <= 255 / > 255), P1 = P2, zero penalties, the largest P1 + P2 the bytes hold."""
    W, H, D, B = 61, 90, 128, 2
    vols = [synth.cost_volume(W, H, D, seed=P2 + f, cmax=cmax) for f in range(B)]
    _, _, off = synth.epi_maps(W, H, "axis")
    with EpiPlan(W, H, D, B, paths=paths) as plan:
        plan.set_penalties(P1, P2, 0.3)
        for f in range(B):
            plan.upload_cost(f, vols[f])
            plan.upload_offset(f, off)
        plan.set_agg_mode(4)
        assert plan.kernel_name == "band16/nowrap"
        plan.run(STAGE_AGGREGATE | STAGE_WTA)
        _check(oracle, plan, vols, off, P1, P2, paths, W, H, D, taps=(0,))


def test_band_sweeps_fall_back_outside_their_range(gpu_lib):
    """Wrapping penalties, P1 > P2, P2 > 127 or a D that is not 16 << k: mode 4 leaves the line kernels in place."""
    with EpiPlan(20, 10, 128, 1, paths=8) as plan:
        plan.set_agg_mode(4)
        plan.set_penalties(6, 64, 0.3)
        assert plan.kernel_name == "band16/nowrap"
        plan.set_penalties(100, 200, 0.3)
        assert plan.kernel_name == "packed16/wrap"
        plan.set_penalties(70, 64, 0.3)
        assert plan.kernel_name == "packed16/nowrap"
        plan.set_penalties(6, 128, 0.3)
        assert plan.kernel_name != "band16/nowrap"
        plan.set_penalties(60, 70, 0.3)                      # P1 + P2 > 127: the biased y no longer fits the byte forms
        assert plan.kernel_name == "packed16/nowrap"
    with EpiPlan(20, 10, 20, 1, paths=8) as plan:
        plan.set_agg_mode(4)
        assert plan.kernel_name == "generic"


def test_band_sweeps_whole_mex_and_batch_copy(gpu_lib, oracle):
    """All stages through the band pipeline (cost stage feeding it), and fsgm_epi_plan_copy_cost's rotated copies."""
    W, H, D, B = 96, 70, 128, 3
    I1, I2 = synth.image_pair(W, H, D, seed=12)
    pd0, nd, off = synth.epi_maps(W, H, "general", seed=13)
    for paths in (8, 4):
        bd, mc = oracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths)
        with EpiPlan(W, H, D, B, paths=paths) as plan:
            plan.set_penalties(6, 64, 0.3)
            for f in range(B):
                plan.upload(f, I1, I2, pd0, nd, off)
            plan.set_agg_mode(4)
            plan.run()
            assert plan.kernel_name == "band16/nowrap"
            for f in range(B):
                gbd, gmc = plan.download(f)
                np.testing.assert_array_equal(gmc, mc)
                np.testing.assert_array_equal(gbd, bd)
    Cv = synth.cost_volume(W, H, D, seed=5, cmax=24)
    with EpiPlan(W, H, D, B, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.upload_cost(0, Cv)
        plan.copy_cost(1, 0, 37)
        plan.copy_cost(2, 0, -5)
        for f in range(B):
            plan.upload_offset(f, off)
        plan.set_agg_mode(4)
        plan.run(STAGE_AGGREGATE | STAGE_WTA)
        np.testing.assert_array_equal(plan.download_cost(1), np.roll(Cv, 37, axis=1))
        np.testing.assert_array_equal(plan.download_cost(2), np.roll(Cv, -5, axis=1))
        _check(oracle, plan, [Cv, np.ascontiguousarray(np.roll(Cv, 37, axis=1)), np.ascontiguousarray(np.roll(Cv, -5, axis=1))],
               off, 6, 64, 8, W, H, D, taps=())


def test_band_sweeps_kitti_shape_against_the_line_kernels_and_the_oracle(gpu_lib, oracle):
    """1242x375x128, 8 and 4 paths: six bands of 64 rows, 1368-step walks; frame 0 against the oracle (S of every voxel),
    every frame against the line kernels on a second plan."""
    W, H, D, B = 1242, 375, 128, 3
    _, _, off = synth.epi_maps(W, H, "axis")
    vols = [synth.cost_volume(W, H, D, seed=900 + f, cmax=24) for f in range(B)]
    for paths in (8, 4):
        with EpiPlan(W, H, D, B, paths=paths) as plan, EpiPlan(W, H, D, B, paths=paths) as ref:
            for pl in (plan, ref):
                pl.set_penalties(6, 64, 0.3)
                for f in range(B):
                    pl.upload_cost(f, vols[f])
                    pl.upload_offset(f, off)
            plan.set_agg_mode(4)
            ref.set_agg_mode(1)
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
            ref.run(STAGE_AGGREGATE | STAGE_WTA)
            assert plan.kernel_name == "band16/nowrap" and ref.kernel_name == "packed16/nowrap"
            for f in range(B):
                a, b = plan.download(f), ref.download(f)
                np.testing.assert_array_equal(a[1], b[1], err_msg=f"{paths} paths, frame {f} minC")
                np.testing.assert_array_equal(a[0], b[0], err_msg=f"{paths} paths, frame {f} bestD")
            S = oracle.epi_aggregate(vols[0], 6, 64, paths)
            np.testing.assert_array_equal(plan.download_sum(0), S[:-1].reshape(H, W, D), err_msg=f"{paths} paths, S")
            bd, mc = oracle.epi_wta(S, W, H, D, 1)
            np.testing.assert_array_equal(plan.download(0)[1], mc)
            np.testing.assert_array_equal(plan.download(0)[0], oracle.epi_vz_to_disp(bd, off, 0.3, D + 1))


CHAIN_SHAPES = [(130, 150, 128, 3), (29, 65, 128, 2), (37, 64, 128, 1), (300, 200, 128, 2), (45, 70, 256, 2), (70, 260, 64, 1), (1, 140, 128, 1), (9, 1, 128, 2)]


@pytest.mark.parametrize("paths", [8, 4])
@pytest.mark.parametrize("W,H,D,B", CHAIN_SHAPES)
def test_chained_band_sweeps_match_the_oracle(gpu_lib, oracle, W, H, D, B, paths):
    """Mode 5: the bands of a frame as workgroups of their own that hand their last row's states over while they run
    (tagged words, ticket order, bounded polls).  Several runs in a row (the hand-off tag changes from launch to launch),
    the S tap in between (it leaves untagged words in a hand-off map), and a switch to the sequential form and back."""
    vols = [synth.cost_volume(W, H, D, seed=W + 7 * H + f, cmax=24) for f in range(B)]
    for v in vols:
        v[:, ::5, :] = 0
    _, _, off = synth.epi_maps(W, H, "general", seed=3)
    with EpiPlan(W, H, D, B, paths=paths) as plan:
        plan.set_penalties(6, 64, 0.3)
        for f in range(B):
            plan.upload_cost(f, vols[f])
            plan.upload_offset(f, off)
        plan.set_agg_mode(5)
        assert plan.kernel_name == "band16chain/nowrap"
        for _ in range(3):
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
        _check(oracle, plan, vols, off, 6, 64, paths, W, H, D, taps=(0,))
        for _ in range(17):                                  # through every tag value and around
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
        _check(oracle, plan, vols, off, 6, 64, paths, W, H, D, taps=())
        plan.set_agg_mode(4)
        plan.run(STAGE_AGGREGATE | STAGE_WTA)
        plan.set_agg_mode(5)
        plan.run(STAGE_AGGREGATE | STAGE_WTA)
        _check(oracle, plan, vols, off, 6, 64, paths, W, H, D, taps=(B - 1,))
        plan.sync()                                          # surfaces a timed-out hand-off, if any


def test_chained_band_sweeps_kitti_shape(gpu_lib, oracle):
    """1242x375x128, 6 bands per frame, 40 frames (240 workgroups in flight, the bands of every frame waiting on one another):
    every frame against the sequential band sweeps, frame 0 against the oracle."""
    W, H, D, B = 1242, 375, 128, 40
    _, _, off = synth.epi_maps(W, H, "axis")
    base = synth.cost_volume(W, H, D, seed=77, cmax=24)
    for paths in (8, 4):
        with EpiPlan(W, H, D, B, paths=paths) as plan:
            plan.set_penalties(6, 64, 0.3)
            plan.upload_cost(0, base)
            plan.upload_offset(0, off)
            for f in range(1, B):
                plan.copy_cost(f, 0, 29 * f)
                plan.upload_offset(f, off)
            plan.set_agg_mode(4)
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
            want = [plan.download(f) for f in range(B)]
            plan.set_agg_mode(5)
            for _ in range(2):
                plan.run(STAGE_AGGREGATE | STAGE_WTA)
            plan.sync()
            for f in range(B):
                got = plan.download(f)
                np.testing.assert_array_equal(got[1], want[f][1], err_msg=f"{paths} paths, frame {f} minC")
                np.testing.assert_array_equal(got[0], want[f][0], err_msg=f"{paths} paths, frame {f} bestD")
            S = oracle.epi_aggregate(base, 6, 64, paths)
            bd, mc = oracle.epi_wta(S, W, H, D, 1)
            np.testing.assert_array_equal(want[0][1], mc)


@pytest.mark.parametrize("paths,mode", [(8, 4), (4, 4), (8, 5)])
def test_band_sweeps_with_more_workgroups_than_the_chip_holds(gpu_lib, oracle, paths, mode):
    """600 frames of 70 x 64 x 128 (mode 4: one workgroup per frame = 600 workgroups; mode 5: 70 rows are two bands, 1200):
    more than two per CU on 256 CUs, so workgroups start as others retire and two share a CU's LDS -- the regime of the
    512-frame headline run, which bench.py only self-checks against the line kernels.  A sample of frames spread over the
    batch against the oracle (minC, bestD; S of two of them), every frame against the first frame that holds the same volume."""
    W, H, D, B = 70, 70 if mode == 5 else 64, 128, 600
    NV = 5                                                   # distinct volumes; frame f holds volume f % NV rolled by f // NV columns
    vols = [synth.cost_volume(W, H, D, seed=4000 + v, cmax=24) for v in range(NV)]
    for v in vols:
        v[:, ::7, :] = 0
    _, _, off = synth.epi_maps(W, H, "general", seed=3)
    with EpiPlan(W, H, D, B, paths=paths) as plan:
        plan.set_penalties(6, 64, 0.3)
        for f in range(B):
            if f < NV:
                plan.upload_cost(f, vols[f])
            else:
                plan.copy_cost(f, f % NV, f // NV)
            plan.upload_offset(f, off)
        plan.set_agg_mode(mode)
        assert plan.kernel_name == ("band16/nowrap" if mode == 4 else "band16chain/nowrap")
        for _ in range(2):
            plan.run(STAGE_AGGREGATE | STAGE_WTA)
        plan.sync()
        sample = sorted({0, 1, 255, 256, 257, 511, 512, 513, B - 1} | {int(round(i * (B - 1) / 11)) for i in range(12)})
        for f in sample:
            v = np.ascontiguousarray(np.roll(vols[f % NV], f // NV, axis=1))
            S = oracle.epi_aggregate(v, 6, 64, paths)
            bd, mc = oracle.epi_wta(S, W, H, D, 1)
            gbd, gmc = plan.download(f)
            np.testing.assert_array_equal(gmc, mc, err_msg=f"frame {f} minC")
            np.testing.assert_array_equal(gbd, oracle.epi_vz_to_disp(bd, off, 0.3, D + 1), err_msg=f"frame {f} bestD")
            if f in (257, B - 1):
                np.testing.assert_array_equal(plan.download_sum(f), S[:-1].reshape(H, W, D), err_msg=f"frame {f} S")
        # frames NV apart in the same residue class hold the same volume rolled by one more column: minC rolls with it only if
        # the roll does not move a path start, so compare exact duplicates instead -- roll by a multiple of W
        dup = [f for f in range(B) if (f // NV) % W == 0 and f >= NV]
        for f in dup:
            a, b = plan.download(f), plan.download(f % NV)
            np.testing.assert_array_equal(a[1], b[1], err_msg=f"frame {f} vs {f % NV} minC")
            np.testing.assert_array_equal(a[0], b[0], err_msg=f"frame {f} vs {f % NV} bestD")
