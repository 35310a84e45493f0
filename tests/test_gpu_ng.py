"""GPU parity tests for the two neighbour-guided variants vs the CPU oracle.
calc_pyd_cost_sgm_ng.cpp:39-78 (sgm_step), :101-299 (sgm2d), :308-368 (subpixel_refine),
:370-446 (calc_cost); calc_cost_sgm_ng.cpp:46-98, :122-186, :188-419."""
import numpy as np
import pytest

from fsgm_amd import synth, calc_pyd_cost_sgm_ng, calc_cost_sgm_ng

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,mvW,mvH,r,agg,sub,P1,P2,kind", [
    (40, 30, 40, 30, 1, 2, 0, 6, 32, "zero"),          # ng_sgm.m:20 argument values
    (37, 23, 30, 20, 1, 5, 1, 6, 32, "general"),       # hint map smaller than the image: clamped (:392-393)
    (33, 21, 40, 25, 1, 2, 1, 90, 120, "general"),     # wrapping penalties
    (21, 15, 21, 15, 2, 3, 0, 6, 32, "even"),          # D = 225
    (12, 1, 12, 1, 1, 2, 0, 6, 32, "general"), (1, 12, 1, 12, 1, 2, 1, 6, 32, "general"),
    (150, 90, 150, 90, 1, 2, 1, 6, 32, "int"),         # several workgroups per direction
])
def test_calc_pyd_cost_sgm_ng_bit_exact(gpu_lib, oracle, W, H, mvW, mvH, r, agg, sub, P1, P2, kind):
    I1, I2 = synth.image_pair(W, H, 16, seed=W + r)
    mv = synth.hint_map(mvW, mvH, kind, seed=H, amp=6.0)
    mc, fl, _, S = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, r, agg, sub, P1, P2, want_volumes=True)
    gmc, gfl, gS = calc_pyd_cost_sgm_ng(I1, I2, mv, r, agg, sub, P1, P2, return_sum=True)
    np.testing.assert_array_equal(gS, S)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gfl, fl)


def test_calc_pyd_cost_sgm_ng_extreme_hints(gpu_lib, oracle):
    """Hints near and beyond the int range: candidate motion vectors of +-2^31 magnitude, for which
    'within 2' must not be decided by a wrapping 32-bit difference (the fast matcher steps aside)."""
    W, H = 48, 20
    I1, I2 = synth.image_pair(W, H, 16, seed=7)
    mv = synth.hint_map(W, H, "int", seed=3, amp=5.0)
    mv[0, 3:9, 10:20] = 2147483646.0
    mv[1, 5:12, 22:30] = -2147483647.0
    mv[0, 10:14, 30:40] = 3.0e9                           # converts to INT_MIN like cvttsd2si
    mv[1, 0:4, 0:6] = 1073741823.0                        # just inside the fast matcher's range
    mc, fl, _, S = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 0, 6, 32, want_volumes=True)
    gmc, gfl, gS = calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 0, 6, 32, return_sum=True)
    np.testing.assert_array_equal(gS, S)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gfl, fl)


@pytest.mark.parametrize("W,H,P1,P2", [(40, 30, 6, 32), (33, 17, 6, 32), (24, 20, 100, 200), (7, 3, 6, 32), (2, 2, 6, 32),
                                        (1, 5, 6, 32), (5, 1, 6, 32), (4, 6, 6, 32), (3, 4, 6, 32), (4, 1, 6, 32), (160, 120, 6, 32)])
def test_calc_cost_sgm_ng_bit_exact(gpu_lib, oracle, W, H, P1, P2):
    I1, I2 = synth.image_pair(W, H, 16, seed=W * H)
    I1 = (I1.astype(np.int32) * 5 % 256).astype(np.uint8)        # strong gradients: adaptive P2 both ways
    rs = oracle.glibc_rand_stream(oracle.sgm_ng_rand_draws(W, H))
    mc, fl = oracle.calc_cost_sgm_ng(I1, I2, P1, P2, rs)
    gmc, gfl = calc_cost_sgm_ng(I1, I2, None, 1, 2, 0, P1, P2, rand_stream=rs)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gfl, fl)


def test_calc_cost_sgm_ng_draws_libc_rand_like_the_reference(gpu_lib, oracle):
    """With no stream given the library consumes libc rand() itself (process-global state, as in a
    MATLAB session): after srand(1) it must reproduce the explicit-stream result."""
    import ctypes
    W, H = 20, 12
    I1, I2 = synth.image_pair(W, H, 16, seed=4)
    rs = oracle.glibc_rand_stream(oracle.sgm_ng_rand_draws(W, H), seed=1)
    want = calc_cost_sgm_ng(I1, I2, None, 1, 2, 0, 6, 32, rand_stream=rs)
    ctypes.CDLL(None).srand(ctypes.c_uint(1))
    got = calc_cost_sgm_ng(I1, I2, None, 1, 2, 0, 6, 32)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])


def test_calc_cost_sgm_ng_exact_matcher_form(gpu_lib, oracle, monkeypatch):
    """The pipelined kernel matches motion vectors as packed 16-bit pairs and switches to the exact
    comparison when one leaves that range -- unreachable on small images, so the exact form is forced here."""
    W, H = 37, 21
    I1, I2 = synth.image_pair(W, H, 16, seed=11)
    rs = oracle.glibc_rand_stream(oracle.sgm_ng_rand_draws(W, H))
    mc, fl = oracle.calc_cost_sgm_ng(I1, I2, 6, 32, rs)
    monkeypatch.setenv("FSGM_OTF_EXACT", "1")
    gmc, gfl = calc_cost_sgm_ng(I1, I2, None, 1, 2, 0, 6, 32, rand_stream=rs)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gfl, fl)


@pytest.mark.parametrize("parts", ["0", "2", "3", "4"])
def test_calc_pyd_cost_sgm_ng_matcher_split(gpu_lib, oracle, monkeypatch, parts):
    """A single frame cuts every candidate's matcher over the predecessor's entries (2 ways by default;
    FSGM_NG_SPLIT selects 0/1 = no split, 2, 3, 4): the fold must keep 'last exact match wins'."""
    W, H = 45, 38
    I1, I2 = synth.image_pair(W, H, 16, seed=6)
    mv = synth.hint_map(W, H, "int", seed=9, amp=2.0)
    mc, fl, _, S = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 1, 6, 32, want_volumes=True)
    monkeypatch.setenv("FSGM_NG_SPLIT", parts)
    gmc, gfl, gS = calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 1, 6, 32, return_sum=True)
    np.testing.assert_array_equal(gS, S)
    np.testing.assert_array_equal(gmc, mc)
    np.testing.assert_array_equal(gfl, fl)


@pytest.mark.parametrize("kind,amp,sub,P1,P2,dedupe", [
    ("zero", 1.0, 0, 6, 32, "1"),          # all nine hints equal: 9 distinct candidates of 81
    ("int", 1.0, 1, 6, 32, "1"),           # neighbouring integer hints: overlapping 3x3 expansions
    ("general", 0.8, 1, 6, 32, "1"),       # fractional hints: equal integer vectors sampled at different places, i.e. different costs
    ("general", 0.8, 0, 90, 120, "1"),     # the same with wrapping penalties
    ("int", 1.0, 1, 6, 32, "0"),           # switch off: every candidate staged
])
def test_calc_pyd_cost_sgm_ng_repeated_candidates(gpu_lib, oracle, monkeypatch, kind, amp, sub, P1, P2, dedupe):
    """The aggregation stages a pixel's candidate list without repeats (same vector and same cost, the last of
    each group kept): results must not change, whatever the share of repeats."""
    W, H = 83, 58
    I1, I2 = synth.image_pair(W, H, 16, seed=13)
    mv = synth.hint_map(W, H, kind, seed=17, amp=amp)
    mc, fl, _, S = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, sub, P1, P2, want_volumes=True)
    monkeypatch.setenv("FSGM_NG_DEDUPE", dedupe)
    for parts in ("2", "0"):                                  # split and one-thread-per-candidate kernels
        monkeypatch.setenv("FSGM_NG_SPLIT", parts)
        gmc, gfl, gS = calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, sub, P1, P2, return_sum=True)
        np.testing.assert_array_equal(gS, S)
        np.testing.assert_array_equal(gmc, mc)
        np.testing.assert_array_equal(gfl, fl)


@pytest.mark.parametrize("kind,amp,sub,P1,P2", [
    ("zero", 1.0, 0, 6, 32),               # one cell cluster: every pixel's vectors in a 3x3 box
    ("int", 1.0, 1, 6, 32),                # boxes of up to 5x5
    ("int", 4.0, 0, 6, 32),                # boxes of up to 11x11: the largest the grid takes, next to pixels that fall back
    ("int", 5.0, 1, 6, 32),                # mostly around the limit (12): box and list steps mixed inside a line
    ("int", 9.0, 0, 6, 32),                # mostly too wide: list steps
    ("general", 3.0, 1, 6, 32),            # fractional hints: one vector at several costs in a cell (the last one counts)
    ("general", 2.0, 0, 90, 120),          # wrapping penalties
])
def test_calc_pyd_cost_sgm_ng_grid_matcher(gpu_lib, oracle, monkeypatch, kind, amp, sub, P1, P2):
    """The grid form of the matcher (ng_agg_grid_kernel: a pixel whose vectors fit a 12x12 box stages its list as a grid,
    the next pixel's candidates read 25 cells) against the oracle, forced on for a single frame, where auto mode would
    take the split list matcher; hint spreads from one cell to far beyond the box."""
    monkeypatch.setenv("FSGM_NG_GRID", "1")
    for W, H in ((83, 58), (17, 40), (200, 9)):
        I1, I2 = synth.image_pair(W, H, 16, seed=W)
        mv = synth.hint_map(W, H, kind, seed=H + 3, amp=amp)
        mc, fl, _, S = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, sub, P1, P2, want_volumes=True)
        gmc, gfl, gS = calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, sub, P1, P2, return_sum=True)
        np.testing.assert_array_equal(gS, S, err_msg=f"{W}x{H}")
        np.testing.assert_array_equal(gmc, mc, err_msg=f"{W}x{H}")
        np.testing.assert_array_equal(gfl, fl, err_msg=f"{W}x{H}")


@pytest.mark.parametrize("kind,amp", [("zero", 1.0), ("int", 3.0), ("general", 1.0)])
def test_calc_pyd_cost_sgm_ng_batch_picks_a_matcher_on_the_device(gpu_lib, oracle, kind, amp):
    """Three frames or more: the list and the grid form of the aggregation are both launched and the mean list length of
    the launch decides on the device which of them runs (short lists: 'zero'; long ones: 'int' with a spread of 7)."""
    from fsgm_amd import calc_pyd_cost_sgm_ng_batch
    W, H = 61, 37
    frames, want = [], []
    for i in range(3):
        I1, I2 = synth.image_pair(W, H, 16, seed=50 + i)
        mv = synth.hint_map(W, H, kind, seed=60 + i, amp=amp)
        frames.append((I1, I2, mv))
        want.append(oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 1, 6, 32))
    for rep in range(2):
        for i, ((gmc, gfl), (mc, fl)) in enumerate(zip(calc_pyd_cost_sgm_ng_batch(frames, 1, 2, 1, 6, 32), want)):
            np.testing.assert_array_equal(gmc, mc, err_msg=f"rep {rep} frame {i}")
            np.testing.assert_array_equal(gfl, fl, err_msg=f"rep {rep} frame {i}")


@pytest.mark.parametrize("kind,amp,sub,P1,P2,n", [
    ("zero", 1.0, 0, 6, 32, 1),            # 9 kept entries of 81
    ("int", 1.0, 1, 6, 32, 3),             # up to 25
    ("int", 2.0, 0, 6, 32, 1),             # up to 49: near the 64 lanes of a wave
    ("int", 3.0, 1, 6, 32, 3),             # some pixels beyond 64 kept entries: the whole launch falls back
    ("general", 0.8, 1, 6, 32, 1),         # fractional hints: equal vectors at different costs stay separate entries
    ("general", 0.8, 0, 90, 120, 3),       # wrapping penalties
])
def test_calc_pyd_cost_sgm_ng_compact_kernel(gpu_lib, oracle, monkeypatch, kind, amp, sub, P1, P2, n):
    """The aggregation over the kept entries only (ng_agg_compact_kernel: one wave per line, sums at the first member of
    every group of repeats, WTA over the groups): S -- read back through launch_ng_fill_repeats --, minC and flow against
    the oracle for lists from 9 entries to beyond what the kernel holds, single frames and batches; and with the kernel
    taken out of the set (FSGM_NG_COMPACT=0) for the same inputs."""
    from fsgm_amd import calc_pyd_cost_sgm_ng_batch
    W, H = 83, 58
    frames, want = [], []
    for i in range(n):
        I1, I2 = synth.image_pair(W, H, 16, seed=13 + i)
        mv = synth.hint_map(W, H, kind, seed=17 + i, amp=amp)
        frames.append((I1, I2, mv))
        want.append(oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, sub, P1, P2, want_volumes=True))
    # "": the kernel's lanes-a-line class picked on the device (one line a wave for a single frame, by the mean list length for
    # a batch); "16" / "32": that class forced, so that lists of up to 64 entries take up to four / two rounds; "0": kernel off
    for compact in ("", "16", "32", "0"):
        if compact == "0":
            monkeypatch.delenv("FSGM_NG_COMPACT_G", raising=False)
            monkeypatch.setenv("FSGM_NG_COMPACT", "0")
        elif compact:
            monkeypatch.setenv("FSGM_NG_COMPACT_G", compact)
        gmc, gfl, gS = calc_pyd_cost_sgm_ng(*frames[0], 1, 2, sub, P1, P2, return_sum=True)
        np.testing.assert_array_equal(gS, want[0][3], err_msg=f"compact {compact!r}")
        np.testing.assert_array_equal(gmc, want[0][0], err_msg=f"compact {compact!r}")
        np.testing.assert_array_equal(gfl, want[0][1], err_msg=f"compact {compact!r}")
        if n > 1:
            for i, (bmc, bfl) in enumerate(calc_pyd_cost_sgm_ng_batch(frames, 1, 2, sub, P1, P2)):
                np.testing.assert_array_equal(bmc, want[i][0], err_msg=f"compact {compact!r} frame {i}")
                np.testing.assert_array_equal(bfl, want[i][1], err_msg=f"compact {compact!r} frame {i}")


def test_ng_batches_match_single_calls(gpu_lib, oracle):
    """Frames of a batch share one launch sequence (three or more frames: one thread per (line, candidate), no
    matcher split; the on-the-fly variant: one workgroup per frame): same results as the oracle frame by frame."""
    from fsgm_amd import calc_pyd_cost_sgm_ng_batch, calc_cost_sgm_ng_batch
    W, H = 52, 31
    frames, want = [], []
    for i in range(3):
        I1, I2 = synth.image_pair(W, H, 16, seed=20 + i)
        mv = synth.hint_map(W, H, ("int", "zero", "general")[i], seed=30 + i, amp=2.0)
        frames.append((I1, I2, mv))
        want.append(oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 1, 6, 32))
    for (gmc, gfl), (mc, fl) in zip(calc_pyd_cost_sgm_ng_batch(frames, 1, 2, 1, 6, 32), want):
        np.testing.assert_array_equal(gmc, mc)
        np.testing.assert_array_equal(gfl, fl)
    W, H = 23, 14
    frames, want = [], []
    for i in range(3):
        I1, I2 = synth.image_pair(W, H, 16, seed=40 + i)
        rs = oracle.glibc_rand_stream(oracle.sgm_ng_rand_draws(W, H), seed=7 + i)
        frames.append((I1, I2, rs))
        want.append(oracle.calc_cost_sgm_ng(I1, I2, 6, 32, rs))
    for (gmc, gfl), (mc, fl) in zip(calc_cost_sgm_ng_batch(frames, 6, 32), want):
        np.testing.assert_array_equal(gmc, mc)
        np.testing.assert_array_equal(gfl, fl)
