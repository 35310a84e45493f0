"""GPU parity tests for the pyramidal driver (pyramidal_sgm.m:1-77 as one device-resident call):
fsgm_pyramidal_sgm / PyramidPlan through the C ABI vs the CPU oracle (oracle/fsgm_oracle_pyramid.cpp).
Everything is compared exactly: gray pyramids (u8), flows of every level (fp64, same IEEE operations
in the same order), minC (u32)."""
import numpy as np
import pytest

from fsgm_amd import synth, pyramidal_sgm, PyramidPlan

pytestmark = pytest.mark.gpu


def _pair(W, H, ch, seed):
    I0, I1 = synth.image_pair(W, H, 12, seed=seed)
    if ch == 3:
        n0 = synth.uniform_u8(seed + 50, (3, H, W), hi=40).astype(np.int32)
        I0 = np.clip(np.stack([I0, I0 // 2 + 60, 255 - I0]).astype(np.int32) + n0 - 20, 0, 255).astype(np.uint8)
        I1 = np.clip(np.stack([I1, I1 // 2 + 60, 255 - I1]).astype(np.int32) + n0 - 20, 0, 255).astype(np.uint8)
    return I0, I1


@pytest.mark.parametrize("W,H,ch,numPyd", [(64, 48, 1, 3), (61, 47, 3, 3), (150, 71, 3, 4), (33, 21, 1, 1),
                                           (40, 30, 1, 5), (97, 17, 3, 2), (2, 3, 1, 2)])
def test_pyramidal_sgm_bit_exact(gpu_lib, oracle, W, H, ch, numPyd):
    I0, I1 = _pair(W, H, ch, seed=W + numPyd)
    want_mv, want_minC, want_lv = oracle.pyramidal_sgm(I0, I1, numPyd)
    mv, mvPyd, minC = pyramidal_sgm(I0, I1, numPyd)
    for l in range(numPyd - 1, -1, -1):                   # coarse to fine: the first difference is the informative one
        np.testing.assert_array_equal(mvPyd[l], want_lv[l], err_msg=f"level {l + 1}")
    np.testing.assert_array_equal(mv, want_mv)
    np.testing.assert_array_equal(minC, want_minC)


def test_pyramid_plans_in_flight(gpu_lib, oracle):
    """Several PyramidPlans started before any is waited for (one stream each, the throughput form of the level loop):
    every plan's flow of every level and minC against the oracle, twice (buffers reused)."""
    W, H, n = 97, 61, 4
    pairs = [_pair(W, H, 3, seed=70 + f) for f in range(n)]
    want = [oracle.pyramidal_sgm(a, b, 3) for a, b in pairs]
    plans = [PyramidPlan(W, H, 3, 3) for _ in range(n)]
    try:
        for pl, (a, b) in zip(plans, pairs):
            pl.upload(a, b)
        for rep in range(2):
            for pl in plans:
                pl.run()
            for pl in plans:
                pl.sync()
            for f, pl in enumerate(plans):
                for l in (3, 2, 1):
                    mv, minC = pl.download(l)
                    np.testing.assert_array_equal(mv, want[f][2][l - 1], err_msg=f"rep {rep} plan {f} level {l}")
                np.testing.assert_array_equal(minC, want[f][1], err_msg=f"rep {rep} plan {f}")
    finally:
        for pl in plans:
            pl.close()


@pytest.mark.parametrize("over", [dict(verSearchHalfWinSize=3, horSearchHalfWinSize=4), dict(adaptiveP2=1, P2=64),
                                  dict(enableDiagonal=0, totalPass=1), dict(aggHalfWinSize=1, P1=10, P2=40),
                                  dict(verSearchHalfWinSize=6, horSearchHalfWinSize=2)])
def test_pyramidal_sgm_parameter_overrides(gpu_lib, oracle, over):
    I0, I1 = _pair(57, 39, 3, seed=11)
    I0 = (I0.astype(np.int32) * 5 % 256).astype(np.uint8)        # strong gradients: adaptive P2 switches
    o = dict(P1=6, P2=32, aggHalfWinSize=2, verSearchHalfWinSize=5, horSearchHalfWinSize=5, enableDiagonal=1, totalPass=2, adaptiveP2=0)
    o.update(over)
    want_mv, want_minC, want_lv = oracle.pyramidal_sgm(I0, I1, 3, o["P1"], o["P2"], o["aggHalfWinSize"], o["verSearchHalfWinSize"],
                                                       o["horSearchHalfWinSize"], o["enableDiagonal"], o["totalPass"], o["adaptiveP2"])
    mv, mvPyd, minC = pyramidal_sgm(I0, I1, 3, **over)
    for l in range(2, -1, -1):
        np.testing.assert_array_equal(mvPyd[l], want_lv[l], err_msg=f"level {l + 1}")
    np.testing.assert_array_equal(minC, want_minC)


def test_pyramid_plan_gray_pyramid_and_reuse(gpu_lib, oracle):
    """impyramid 'reduce' + rgb2gray on the device vs the oracle, level by level; the plan is reused for a
    second pair (nothing of the first run may leak into it)."""
    W, H, n = 131, 77, 4
    with PyramidPlan(W, H, 3, n) as plan:
        for seed in (5, 6):
            I0, I1 = _pair(W, H, 3, seed=seed)
            plan.upload(I0, I1)
            plan.run()
            c0, c1 = I0, I1
            for l in range(1, n + 1):
                if l > 1:
                    c0 = np.stack([oracle.impyramid_reduce(c) for c in c0])
                    c1 = np.stack([oracle.impyramid_reduce(c) for c in c1])
                g0, g1 = plan.download_gray(l)
                np.testing.assert_array_equal(g0, oracle.rgb2gray(c0), err_msg=f"level {l}")
                np.testing.assert_array_equal(g1, oracle.rgb2gray(c1), err_msg=f"level {l}")
                assert plan.level_size(l) == (c0.shape[2], c0.shape[1])
            want_mv, want_minC, _ = oracle.pyramidal_sgm(I0, I1, n)
            mv, minC = plan.download(1)
            np.testing.assert_array_equal(mv, want_mv)
            np.testing.assert_array_equal(minC, want_minC)


def test_pyramidal_sgm_kitti_shape_level_sizes(gpu_lib, oracle):
    """BASELINE config 4 shape: 1242x375, 3 levels -> 621x188, 311x94 (ceil(size/2), test_psgm.m:33); every level's
    flow and the finest level's minC against the oracle's pyramidal_sgm (pyramidal_sgm.m:24-76 around
    calc_pyd_cost_sgm.cpp:114-372), all pixels -- levels 1 and 2 included (~35 s of oracle)."""
    with PyramidPlan(1242, 375, 3, 3) as plan:
        assert [plan.level_size(l) for l in (1, 2, 3)] == [(1242, 375), (621, 188), (311, 94)]
        I0, I1 = _pair(1242, 375, 3, seed=1)
        plan.upload(I0, I1)
        plan.run()
        mv, minC = plan.download(1)
        assert mv.shape == (2, 375, 1242) and np.isfinite(mv).all()
        # the coarsest level on its own, from the oracle's pieces
        c0, c1 = I0, I1
        for _ in range(2):
            c0 = np.stack([oracle.impyramid_reduce(c) for c in c0])
            c1 = np.stack([oracle.impyramid_reduce(c) for c in c1])
        g0, g1 = oracle.rgb2gray(c0), oracle.rgb2gray(c1)
        bd, mc, ms = oracle.calc_pyd_cost_sgm(g0, g1, np.zeros((2, 94, 311)), 5, 5, 2, 0, 6, 32, 1, 2, 0)
        lv3, mc3 = plan.download(3)
        np.testing.assert_array_equal(mc3, mc)
        np.testing.assert_array_equal(lv3[0], (bd // 11).astype(np.float64) - 5)
        np.testing.assert_array_equal(lv3[1], (bd % 11).astype(np.float64) - 5)
        # the whole loop: levels 3, 2 and 1 exactly
        want_mv, want_minC, want_lv = oracle.pyramidal_sgm(I0, I1, 3)
        for l in (3, 2, 1):
            got, _ = plan.download(l)
            np.testing.assert_array_equal(got, want_lv[l - 1], err_msg=f"level {l} flow")
        np.testing.assert_array_equal(minC, want_minC)
        np.testing.assert_array_equal(mv, want_mv)
        assert np.abs(mv).max() > 2                             # the hints did travel down the pyramid


@pytest.mark.parametrize("W,H,ch,numPyd,sub", [(61, 45, 3, 3, 0), (48, 37, 1, 2, 1), (5, 4, 1, 2, 0), (33, 21, 1, 1, 1), (2, 3, 3, 3, 0)])
def test_pyramidal_loop_with_the_neighbour_guided_matcher(gpu_lib, oracle, W, H, ch, numPyd, sub):
    """BASELINE config 4 names calc_pyd_cost_sgm_ng for the pyramidal path: the level loop with the ng MEX swapped
    in (fsgm_amd.pyramidal_sgm_ng) against the same composition of the oracle's functions, level by level."""
    from fsgm_amd import pyramidal_sgm_ng
    g0, g1 = synth.image_pair(W, H, 8, seed=W)
    if ch == 3:
        I0 = np.stack([g0, 255 - g0, g0 // 2 + 40]); I1 = np.stack([g1, 255 - g1, g1 // 2 + 40])
    else:
        I0, I1 = g0, g1
    flow, flows, minC = pyramidal_sgm_ng(I0, I1, numPyd, subPixelRefine=sub)
    # oracle composition
    lv = [(I0, I1)]
    for _ in range(1, numPyd):
        a, b = lv[-1]
        red = (lambda im: np.stack([oracle.impyramid_reduce(c) for c in im])) if ch == 3 else oracle.impyramid_reduce
        lv.append((red(a), red(b)))
    gray = [(oracle.rgb2gray(a), oracle.rgb2gray(b)) if ch == 3 else (a, b) for a, b in lv]
    hc, wc = gray[-1][0].shape
    mvPre = np.zeros((2, hc, wc))
    want = []
    for l in range(numPyd, 0, -1):
        mc, fl = oracle.calc_pyd_cost_sgm_ng(gray[l - 1][0], gray[l - 1][1], mvPre, 1, 2, sub, 6, 32)
        want.append(fl)
        mvPre = np.ascontiguousarray(2.0 * np.repeat(np.repeat(fl, 2, axis=1), 2, axis=2))
    assert len(flows) == numPyd
    for got, w in zip(flows, want):
        np.testing.assert_array_equal(got, w)
    np.testing.assert_array_equal(minC, mc)
    np.testing.assert_array_equal(flow, want[-1])


def test_pyramidal_loop_ng_with_a_wider_search(gpu_lib, oracle):
    """halfSearchWinSize = 2: 225 candidates per pixel, the generic aggregation kernel (no repeat removal), P2 = 64."""
    from fsgm_amd import pyramidal_sgm_ng
    W, H = 29, 22
    I0, I1 = synth.image_pair(W, H, 8, seed=3)
    flow, flows, minC = pyramidal_sgm_ng(I0, I1, 2, halfSearchWinSize=2, aggSize=3, P2=64)
    g = [(I0, I1), (oracle.impyramid_reduce(I0), oracle.impyramid_reduce(I1))]
    mvPre = np.zeros((2,) + g[1][0].shape)
    for l in (2, 1):
        mc, fl = oracle.calc_pyd_cost_sgm_ng(g[l - 1][0], g[l - 1][1], mvPre, 2, 3, 0, 6, 64)
        np.testing.assert_array_equal(flows[2 - l], fl)
        mvPre = np.ascontiguousarray(2.0 * np.repeat(np.repeat(fl, 2, axis=1), 2, axis=2))
    np.testing.assert_array_equal(minC, mc)
    np.testing.assert_array_equal(flow, fl)


def test_ng_pyramid_batch_matches_single_pairs(gpu_lib, oracle):
    """A batch of image pairs through one NgPyramidPlan (every kernel of a level covers all frames) = the same pairs one
    at a time = the oracle composition (frame 0), all levels."""
    from fsgm_amd import NgPyramidPlan, pyramidal_sgm_ng
    W, H, B = 83, 61, 5
    pairs = []
    for f in range(B):
        g0, g1 = synth.image_pair(W, H, 8, seed=11 + f)
        pairs.append((np.stack([g0, 255 - g0, g0 // 2 + 40]), np.stack([g1, 255 - g1, g1 // 2 + 40])))
    with NgPyramidPlan(W, H, 3, 3, batch=B) as plan:
        for f, (a, b) in enumerate(pairs):
            plan.upload(a, b, frame=f)
        for _ in range(2):                                      # twice: buffers are reused
            plan.run()
        got = [[plan.download(l, frame=f) for l in (3, 2, 1)] for f in range(B)]
    for f, (a, b) in enumerate(pairs):
        flow, flows, minC = pyramidal_sgm_ng(a, b, 3)
        for li in range(3):
            np.testing.assert_array_equal(got[f][li][0], flows[li], err_msg=f"frame {f} level {3 - li}")
        np.testing.assert_array_equal(got[f][2][1], minC, err_msg=f"frame {f}")


def test_pyramid_batch_matches_single_pairs(gpu_lib, oracle):
    """A batch of image pairs through one PyramidPlan (every kernel of a level covers all frames: fsgm_pyramid_plan_create_batch)
    = the same pairs one at a time = the oracle's pyramidal_sgm (pyramidal_sgm.m:24-76), all levels, RGB and gray."""
    W, H, B = 83, 61, 5
    for ch in (3, 1):
        pairs = [_pair(W, H, ch, seed=21 + f) for f in range(B)]
        with PyramidPlan(W, H, ch, 3, batch=B) as plan:
            for f, (a, b) in enumerate(pairs):
                plan.upload(a, b, frame=f)
            for _ in range(2):                                  # twice: buffers are reused
                plan.run()
            got = [[plan.download(l, frame=f) for l in (1, 2, 3)] for f in range(B)]
            # the gray levels of frames other than 0 (impyramid 'reduce' / rgb2gray per frame of the batch)
            for f in (1, B - 1):
                c0, c1 = pairs[f]
                for l in (1, 2, 3):
                    if l > 1:
                        c0 = np.stack([oracle.impyramid_reduce(c) for c in c0]) if ch == 3 else oracle.impyramid_reduce(c0)
                        c1 = np.stack([oracle.impyramid_reduce(c) for c in c1]) if ch == 3 else oracle.impyramid_reduce(c1)
                    g0, g1 = plan.download_gray(l, frame=f)
                    np.testing.assert_array_equal(g0, oracle.rgb2gray(c0) if ch == 3 else c0, err_msg=f"{ch} channels, frame {f}, gray level {l}")
                    np.testing.assert_array_equal(g1, oracle.rgb2gray(c1) if ch == 3 else c1, err_msg=f"{ch} channels, frame {f}, gray level {l}")
        for f, (a, b) in enumerate(pairs):
            want_mv, want_minC, want_lv = oracle.pyramidal_sgm(a, b, 3)
            for l in (1, 2, 3):
                np.testing.assert_array_equal(got[f][l - 1][0], want_lv[l - 1], err_msg=f"{ch} channels, frame {f}, level {l}")
            np.testing.assert_array_equal(got[f][0][1], want_minC, err_msg=f"{ch} channels, frame {f}")


def test_host_calls_overwrite_a_previous_result(gpu_lib):
    """out= of pyramidal_sgm / pyramidal_sgm_ng: a previous call's result tuple is filled in place (what a caller in a loop does to
    keep its output pages resident); a tuple of another shape is refused."""
    from fsgm_amd import pyramidal_sgm_ng
    W, H = 70, 45
    a0, a1 = _pair(W, H, 3, seed=5)
    b0, b1 = _pair(W, H, 3, seed=6)
    for fn in (pyramidal_sgm, pyramidal_sgm_ng):
        first = fn(a0, a1, 3)
        want = fn(b0, b1, 3)
        again = fn(b0, b1, 3, out=first)
        assert again[0] is first[0] and again[2] is first[2] and all(x is y for x, y in zip(again[1], first[1]))
        np.testing.assert_array_equal(again[0], want[0])
        np.testing.assert_array_equal(again[2], want[2])
        for x, y in zip(again[1], want[1]):
            np.testing.assert_array_equal(x, y)
        with pytest.raises(ValueError):
            fn(b0[:, :-1], b1[:, :-1], 3, out=first)
        with pytest.raises(ValueError):
            fn(b0, b1, 2, out=first)
