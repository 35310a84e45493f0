"""CPU-side tests of the C-ABI library and the host wrappers (no compute calls: this box has no GPU)."""
import ctypes
import os
import re
import numpy as np
import pytest

import fsgm_amd
from fsgm_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = set()
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        txt = open(os.path.join(ROOT, "include", fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(fsgm_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported by libfsgm_hip.so: {missing}"


def test_library_is_gfx950_code_object():
    """The shared object must carry a gfx950 device code object (built by hipcc --offload-arch=gfx950)."""
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"agg_packed_kernel" in blob


def test_no_cpu_fallback_without_device():
    lib = _lib.load()
    if lib.fsgm_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(fsgm_amd.FsgmError) as ei:
        fsgm_amd.EpiPlan(64, 48, 16)
    assert ei.value.status == 2 and "no HIP device" in str(ei.value)
    I1, I2 = synth.image_pair(32, 24, 16)
    pd0, nd, off = synth.epi_maps(32, 24)
    with pytest.raises(fsgm_amd.FsgmError):
        fsgm_amd.calc_cost_sgm(I1, I2, 16, 0.3, pd0, nd, off, 6, 64)


def test_default_params_are_the_shipped_switches():
    p = _lib.load().fsgm_epi_params_default()
    assert (p.paths, p.subpixel, p.vz_to_disp, p.device) == (4, 1, 1, 0)     # calc_cost_sgm.cpp:104,560,4


def test_argument_validation_is_loud():
    I1, I2 = synth.image_pair(32, 24, 16)
    pd0, nd, off = synth.epi_maps(32, 24)
    with pytest.raises(TypeError):
        fsgm_amd.calc_cost_sgm(I1.astype(np.float32), I2, 16, 0.3, pd0, nd, off, 6, 64)
    with pytest.raises(TypeError):
        fsgm_amd.calc_cost_sgm(I1, I2, 16, 0.3, pd0[0], nd, off, 6, 64)
    with pytest.raises(TypeError):
        fsgm_amd.calc_cost_sgm(I1, I2, 16, 0.3, pd0, nd, off.astype(np.float32), 6, 64)
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.fsgm_epi_plan_create(ctypes.byref(h), 0, 10, 16, 1, None) == 1          # FSGM_ERR_INVALID
    assert b"width/height" in lib.fsgm_last_error()
    prm = lib.fsgm_epi_params_default()
    prm.paths = 5
    assert lib.fsgm_epi_plan_create(ctypes.byref(h), 10, 10, 16, 1, ctypes.byref(prm)) == 1
    assert lib.fsgm_epi_plan_create(ctypes.byref(h), 10, 10, 5000, 1, None) == 4       # FSGM_ERR_UNSUPPORTED
    assert lib.fsgm_epi_plan_run(None, 7) == 1


def test_synthetic_inputs_are_deterministic():
    I1, I2 = synth.image_pair(64, 48, 16, seed=1)
    assert int(I1.astype(np.uint64).sum()) > 0
    a = synth.cost_volume(8, 4, 16, seed=3)
    b = synth.cost_volume(8, 4, 16, seed=3)
    np.testing.assert_array_equal(a, b)
    assert a.max() <= 24
    assert synth.splitmix64(1, 3).tolist() == synth.splitmix64(1, 3).tolist()
    assert int(synth.splitmix64(0, 1)[0]) == 0xE220A8397B1DCDAF                 # splitmix64(0) known answer


def test_window_mean_rounding_identity_of_the_rows_cost_kernel():
    """pyd_rows_cost_kernel replaces (u8)(1.0*sum/win + 0.5) (calc_pyd_cost_sgm.cpp:431, fp64) by
    ((2*sum + win) * inv) >> 20 with inv = 2^20 / (2*win) + 1: equal for every sum a window can produce
    (win = 1, 9, 25 taps of at most 32 each)."""
    for win in (1, 9, 25):
        inv = (1 << 20) // (2 * win) + 1
        s = np.arange(0, 32 * win + 1, dtype=np.int64)
        want = (1.0 * s.astype(np.float64) / win + 0.5).astype(np.int64)
        got = ((2 * s + win) * inv) >> 20
        np.testing.assert_array_equal(got, want)
        assert ((2 * s + win) * inv).max() < 2 ** 32


def test_auto_mode_table():
    """fsgm_epi_auto_pipeline (a pure function: no device).  KITTI shape, 8 paths, no-wrap penalties: line kernels below 4 frames,
    parallel sweeps below 26 (meeting in the middle from 10 on), the block sweep pipeline from there, the band sweeps where a round of them pays (230..256 frames,
    473..512), the chained form between the rounds; 4 paths: line kernels below 9 frames, then the pair kernels, then bands.
    Smaller frames: the switch points move with voxels^(-2/3) -- measured at 320x240x64 (profiles/r03_crossover_320x240x64.txt):
    line kernels fastest up to ~26 frames at 8 paths and ~48 at 4, block sweeps from ~80, band sweeps at 4 paths / 512 frames only."""
    from fsgm_amd import auto_pipeline
    kitti = [(8, 1, "packed16/nowrap"), (8, 3, "packed16/nowrap"), (8, 4, "sweep16par/nowrap"), (8, 9, "sweep16par/nowrap"), (8, 10, "sweep16mid/nowrap"),
             (8, 25, "sweep16mid/nowrap"), (8, 26, "sweep16/nowrap"),
             (8, 200, "sweep16/nowrap"), (8, 256, "band16/nowrap"), (8, 300, "band16chain/nowrap"), (8, 512, "band16/nowrap"),
             (4, 8, "packed16/nowrap"), (4, 9, "pairs16/nowrap"), (4, 128, "pairs16/nowrap"), (4, 512, "band16/nowrap")]
    for paths, B, name in kitti:
        assert auto_pipeline(1242, 375, 128, B, paths, 6, 64) == name, (paths, B)
    small = [(8, 16, "packed16/nowrap"), (8, 40, "sweep16par/nowrap"), (8, 128, "sweep16mid/nowrap"), (8, 140, "sweep16/nowrap"), (8, 512, "sweep16/nowrap"),
             (4, 40, "packed16/nowrap"), (4, 128, "pairs16/nowrap"), (4, 512, "band16/nowrap")]
    for paths, B, name in small:
        assert auto_pipeline(320, 240, 64, B, paths, 6, 64) == name, (paths, B)
    assert auto_pipeline(32, 16, 64, 200, 8, 6, 64) == "packed16/nowrap"          # tiny frames: nothing to fuse for
    assert auto_pipeline(1242, 375, 128, 512, 8, 100, 200) == "packed16/wrap"      # wrapping penalties: the exact u8 line kernels
    assert auto_pipeline(1242, 375, 128, 512, 8, 64, 6) == "packed16/nowrap"       # P1 > P2: not a fused pipeline's case
    assert auto_pipeline(1242, 375, 100, 512, 8, 6, 64) == "generic"               # dMax not a multiple of 16
    assert auto_pipeline(0, 375, 128, 1, 8, 6, 64) == ""


def test_round_half_away_as_one_addition_of_pred_half():
    """epi_cost.hip / fsgm_device.h round_clamp_small: max(trunc(v + pred(0.5)), 0) == max(C round(v), 0) for |v| < 2^30,
    checked on both sides of every half integer and integer (the doubles next to them), across magnitudes, and at random."""
    h = np.nextafter(0.5, 0.0)

    def c_round_clamped(v):
        a = np.abs(v)
        r = np.floor(a) + ((a - np.floor(a)) >= 0.5)
        return np.maximum(np.where(v < 0, -r, r), 0.0)

    ks = np.concatenate([np.arange(0, 4096), 2.0 ** np.arange(12, 31), 2.0 ** np.arange(12, 31) - 1, [2.0 ** 30 - 7]])
    vs = []
    for base in (ks + 0.5, ks):
        up, dn = base.copy(), base.copy()
        vs.append(base)
        for _ in range(5):
            up, dn = np.nextafter(up, np.inf), np.nextafter(dn, -np.inf)
            vs += [up, dn]
    vs.append(np.array([0.0, -0.0, 5e-324, -5e-324, h, -h, np.nextafter(h, 0), 0.25, -0.25, -0.5, np.nextafter(-0.5, 0), np.nextafter(-0.5, -1)]))
    v = np.concatenate(vs)
    v = np.concatenate([v, -v])
    np.testing.assert_array_equal(np.maximum(np.trunc(v + h), 0.0), c_round_clamped(v))
    assert np.trunc(h + 0.5) == 1.0                       # what the plain + 0.5 gets wrong
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.uniform(-200, 3000, 2_000_000),
                        np.round(rng.uniform(0, 2 ** 30, 1_000_000)) + rng.choice([0.5, -0.5, 0.49999999, 0.50000001], 1_000_000)])
    np.testing.assert_array_equal(np.maximum(np.trunc(v + h), 0.0), c_round_clamped(v))


def test_box_mean_as_one_fp16_multiply():
    """epi_cost.hip: (u8)(1.0 * s / 25 + 0.5) (calc_cost_sgm.cpp:403-404) == (2 s + 25) / 50 == the u16 pattern of s, read as a
    denormal fp16, times fp16(0.04) -- for every sum below 1024 (a 5x5 window of census costs reaches 600)."""
    s = np.arange(0, 1024, dtype=np.uint16)
    got = (s.view(np.float16) * np.float16(0.04)).view(np.uint16)
    assert np.float16(0.04).view(np.uint16) == 0x291F
    np.testing.assert_array_equal(got, ((2 * s.astype(np.int64) + 25) // 50).astype(np.uint16))
    np.testing.assert_array_equal(got[:601], (1.0 * s[:601] / 25 + 0.5).astype(np.uint8))


def test_device_list_helpers():
    """fsgm_parse_device_list / fsgm_shard_frames (include/fsgm.h "Device lists"): the text format of FSGM_DEVICES and the
    partition frame i -> entry i mod n -- the same round-robin as fsgm_amd.batch.shard_indices uses across processes."""
    import ctypes as C
    from fsgm_amd.batch import shard_indices
    lib = _lib.load()
    buf = (C.c_int32 * 16)()
    for text, want in (("0,1,2", [0, 1, 2]), (" 3 ; 1,,7 ", [3, 1, 7]), ("0", [0]), ("", []), ("0,0", [0, 0])):
        n = lib.fsgm_parse_device_list(text.encode(), buf, 16)
        assert n == len(want) and list(buf[:n]) == want
    for text in ("0,x", "-1", "1.5", "a"):
        assert lib.fsgm_parse_device_list(text.encode(), buf, 16) == -1
    assert lib.fsgm_parse_device_list(b"0,1,2,3", buf, 2) == 2              # never writes past max_devices
    for n_frames, n_dev in ((8, 2), (8, 8), (7, 3), (3, 5), (0, 2), (1, 1)):
        seen = []
        for slot in range(n_dev):
            out, cnt = (C.c_int32 * 64)(), C.c_int32()
            lib.fsgm_shard_frames(n_frames, n_dev, slot, out, C.byref(cnt))
            assert list(out[:cnt.value]) == shard_indices(n_frames, slot, n_dev)
            seen += list(out[:cnt.value])
        assert sorted(seen) == list(range(n_frames))                         # a partition: every frame exactly once
    # without a device the multi-device calls fail loudly too (no CPU fallback)
    if lib.fsgm_device_count() == 0:
        from fsgm_amd import calc_cost_sgm_batch
        I1, I2 = synth.image_pair(16, 8, 8, seed=1)
        maps = synth.epi_maps(16, 8, "axis")
        with pytest.raises(fsgm_amd.FsgmError):
            calc_cost_sgm_batch([(I1, I2) + maps] * 2, 8, 0.3, 6, 64, devices=[0, 0])
