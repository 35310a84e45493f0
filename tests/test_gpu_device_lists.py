"""The in-process multi-device batch calls (include/fsgm.h "Device lists"): frame i on entry i mod n of a
device list, one host thread per entry, no collective.  On the one-GPU test box the list is {0, 0} (and longer): entries are
taken modulo the device count, so every entry is device 0 and the slots take turns -- the partition, the threads, the plan
caches per device and the result scatter are exercised as on a node of 8; concurrency across GPUs is not (unmeasured on
hardware: no multi-GPU box is available to a round).  Results must equal single calls and the oracle bit for bit.
"""
import os

import numpy as np
import pytest

import fsgm_amd
from fsgm_amd import synth
from tests import mexharness as mh

pytestmark = pytest.mark.gpu


def _epi_frames(W, H, D, n):
    out = []
    for s in range(n):
        I1, I2 = synth.image_pair(W, H, D, seed=30 + s)
        out.append((I1, I2) + synth.epi_maps(W, H, "general", seed=50 + s))
    return out


@pytest.mark.parametrize("paths", [4, 8])
@pytest.mark.parametrize("devices", [[0, 0], "0,0,0", [0, 1, 2, 3, 4, 5, 6, 7], [5]])
def test_epi_batch_over_a_device_list(gpu_lib, oracle, paths, devices):
    W, H, D, n = 83, 47, 64, 8
    frames = _epi_frames(W, H, D, n)
    got = fsgm_amd.calc_cost_sgm_batch(frames, D, 0.3, 6, 64, paths=paths, devices=devices)
    assert len(got) == n
    for f, (bd, mc) in zip(frames, got):
        sbd, smc = fsgm_amd.calc_cost_sgm(*f[:2], D, 0.3, *f[2:], 6, 64, paths=paths)
        np.testing.assert_array_equal(bd, sbd)
        np.testing.assert_array_equal(mc, smc)
    for i in (0, 3, 7):
        rbd, rmc = oracle.calc_cost_sgm(*frames[i][:2], D, 0.3, *frames[i][2:], 6, 64, paths)
        np.testing.assert_array_equal(got[i][0], rbd)
        np.testing.assert_array_equal(got[i][1], rmc)


def test_fewer_frames_than_list_entries_and_bad_lists(gpu_lib, oracle):
    W, H, D = 40, 30, 16
    frames = _epi_frames(W, H, D, 3)
    got = fsgm_amd.calc_cost_sgm_batch(frames, D, 0.3, 6, 64, devices=[0, 0, 0, 0, 0])     # two entries stay idle
    for f, (bd, mc) in zip(frames, got):
        rbd, rmc = oracle.calc_cost_sgm(*f[:2], D, 0.3, *f[2:], 6, 64, 4)
        np.testing.assert_array_equal(bd, rbd)
        np.testing.assert_array_equal(mc, rmc)
    with pytest.raises(fsgm_amd.FsgmError, match="negative"):
        fsgm_amd.calc_cost_sgm_batch(frames, D, 0.3, 6, 64, devices=[0, -1])
    with pytest.raises(ValueError):
        fsgm_amd.calc_cost_sgm_batch(frames, D, 0.3, 6, 64, devices="0,x")
    # an entry that fails reports which one
    bad = [frames[0], (frames[1][0], frames[1][1], frames[1][2], frames[1][3], frames[1][4])]
    with pytest.raises(fsgm_amd.FsgmError, match="device list entry"):
        fsgm_amd.calc_cost_sgm_batch(bad, 4096, 0.3, 6, 64, devices=[0, 0])                # dMax beyond the supported maximum


def test_pyd_and_ng_batches_over_a_device_list(gpu_lib, oracle):
    W, H = 52, 34
    pyd_frames = []
    for s in range(5):
        I1, I2 = synth.image_pair(W, H, 16, seed=60 + s)
        pyd_frames.append((I1, I2, synth.hint_map(W, H, "general", seed=s)))
    got = fsgm_amd.calc_pyd_cost_sgm_batch(pyd_frames, 3, 2, 2, 1, 6, 32, 1, 2, 0, devices=[0, 0])
    for (I1, I2, mv), (bd, mc, ms) in zip(pyd_frames, got):
        rbd, rmc, rms = oracle.calc_pyd_cost_sgm(I1, I2, mv, 3, 2, 2, 1, 6, 32, 1, 2, 0)
        np.testing.assert_array_equal(bd, rbd)
        np.testing.assert_array_equal(mc, rmc)
        np.testing.assert_array_equal(ms, rms)
    got = fsgm_amd.calc_pyd_cost_sgm_ng_batch(pyd_frames, 1, 2, 0, 6, 32, devices=[0, 0, 0])
    for (I1, I2, mv), (mc, fl) in zip(pyd_frames, got):
        rmc, rfl = oracle.calc_pyd_cost_sgm_ng(I1, I2, mv, 1, 2, 0, 6, 32)
        np.testing.assert_array_equal(mc, rmc)
        np.testing.assert_array_equal(fl, rfl)
    # the on-the-fly variant: explicit rand() streams per frame
    w, h = 20, 12
    otf = []
    for s in range(3):
        I1, I2 = synth.image_pair(w, h, 16, seed=70 + s)
        otf.append((I1, I2, oracle.glibc_rand_stream(oracle.sgm_ng_rand_draws(w, h), seed=1 + s)))
    got = fsgm_amd.calc_cost_sgm_ng_batch(otf, 6, 32, devices=[0, 0])
    for (I1, I2, rs), (mc, fl) in zip(otf, got):
        rmc, rfl = oracle.calc_cost_sgm_ng(I1, I2, 6, 32, rs)
        np.testing.assert_array_equal(mc, rmc)
        np.testing.assert_array_equal(fl, rfl)


def test_pyramid_drivers_over_a_device_list(gpu_lib, oracle):
    W, H = 70, 46
    pairs = [synth.image_pair(W, H, 16, seed=80 + s) for s in range(3)]
    got = fsgm_amd.pyramidal_sgm_batch(pairs, 2, devices=[0, 0])
    for (I0, I1), (mv, lv, mc) in zip(pairs, got):
        smv, slv, smc = fsgm_amd.pyramidal_sgm(I0, I1, 2)
        np.testing.assert_array_equal(mv, smv)
        np.testing.assert_array_equal(mc, smc)
        for a, b in zip(lv, slv):
            np.testing.assert_array_equal(a, b)
    got = fsgm_amd.pyramidal_sgm_ng_batch(pairs, 2, devices=[0, 0])
    for (I0, I1), (fl, lv, mc) in zip(pairs, got):
        sfl, slv, smc = fsgm_amd.pyramidal_sgm_ng(I0, I1, 2)
        np.testing.assert_array_equal(fl, sfl)
        np.testing.assert_array_equal(mc, smc)


def test_gateway_batch_with_FSGM_DEVICES(gpu_lib, oracle):
    """calc_cost_sgm handed W x H x n arrays: n independent calls of the reference MEX, spread over FSGM_DEVICES."""
    W, H, D, n = 61, 37, 32, 4
    frames = _epi_frames(W, H, D, n)
    I1 = np.stack([f[0] for f in frames]); I2 = np.stack([f[1] for f in frames])
    pd0 = np.stack([f[2] for f in frames]); nd = np.stack([f[3] for f in frames]); off = np.stack([f[4] for f in frames])
    os.environ["FSGM_DEVICES"] = "0,0"
    try:
        (bestD, minC), _ = mh.call("calc_cost_sgm", 2, I1, I2, D, 0.3, pd0, nd, off, 6, 64)
    finally:
        del os.environ["FSGM_DEVICES"]
    assert bestD.shape == (n, H, W) and minC.shape == (n, H, W)
    for i, f in enumerate(frames):
        rbd, rmc = oracle.calc_cost_sgm(*f[:2], D, 0.3, *f[2:], 6, 64, 4)
        np.testing.assert_array_equal(bestD[i], rbd)
        np.testing.assert_array_equal(minC[i], rmc)
    (only,), _ = mh.call("calc_cost_sgm", 1, I1, I2, D, 0.3, pd0, nd, off, 6, 64)         # FSGM_DEVICES unset: FSGM_DEVICE
    np.testing.assert_array_equal(only, bestD)
    with pytest.raises(mh.MexError, match="fsgm:size"):
        mh.call("calc_cost_sgm", 2, I1, I2, D, 0.3, pd0[:2], nd, off, 6, 64)
