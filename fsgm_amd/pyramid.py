"""Host-side mirror of the reference's pyramidal driver pyramidal_sgm.m (the MATLAB function around the
calc_pyd_cost_sgm MEX), on top of the C ABI: the whole level loop runs on the device.

Arrays follow the memory order the reference's drivers hand a MEX (after permute([2 1 3])): images
(height, width) or (3, height, width) uint8 C-contiguous, flows (2, height, width) float64, plane 0 = x.
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import check, ptr


class PyramidParams(C.Structure):
    _fields_ = [("numPyd", C.c_int32), ("P1", C.c_int32), ("P2", C.c_int32), ("aggHalfWinSize", C.c_int32),
                ("verSearchHalfWinSize", C.c_int32), ("horSearchHalfWinSize", C.c_int32), ("enableDiagonal", C.c_int32),
                ("totalPass", C.c_int32), ("adaptiveP2", C.c_int32), ("device", C.c_int32)]


def _bind(lib):
    if getattr(lib, "_pyramid_bound", False):
        return
    vp, i32 = C.c_void_p, C.c_int32
    lib.fsgm_pyramid_params_default.restype = PyramidParams
    lib.fsgm_pyramidal_sgm_host.argtypes = [vp, vp, i32, i32, i32, C.POINTER(PyramidParams), vp, vp, vp]
    lib.fsgm_pyramid_plan_create.argtypes = [C.POINTER(vp), i32, i32, i32, C.POINTER(PyramidParams)]
    lib.fsgm_pyramid_plan_create_batch.argtypes = [C.POINTER(vp), i32, i32, i32, C.POINTER(PyramidParams), i32]
    lib.fsgm_pyramid_plan_upload_frame.argtypes = [vp, i32, vp, vp]
    lib.fsgm_pyramid_plan_download_frame.argtypes = [vp, i32, i32, vp, vp]
    lib.fsgm_pyramid_plan_destroy.argtypes = [vp]
    lib.fsgm_pyramid_plan_destroy.restype = None
    lib.fsgm_pyramid_plan_level_size.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.fsgm_pyramid_plan_upload.argtypes = [vp, vp, vp]
    lib.fsgm_pyramid_plan_run.argtypes = [vp]
    lib.fsgm_pyramid_plan_run_images.argtypes = [vp]
    lib.fsgm_pyramid_plan_sync.argtypes = [vp]
    lib.fsgm_pyramid_plan_download.argtypes = [vp, i32, vp, vp]
    lib.fsgm_pyramid_plan_download_gray.argtypes = [vp, i32, vp, vp]
    lib.fsgm_pyramid_plan_download_gray_frame.argtypes = [vp, i32, i32, vp, vp]
    lib.fsgm_pyramid_plan_time.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
    lib._pyramid_bound = True


def _params(lib, numPyd, device, overrides):
    prm = lib.fsgm_pyramid_params_default()
    prm.numPyd, prm.device = int(numPyd), int(device)
    for k, v in overrides.items():
        if not hasattr(prm, k):
            raise TypeError(f"unknown pyramidal_sgm parameter {k!r}")
        setattr(prm, k, int(v))
    return prm


def _check_images(I0, I1):
    I0, I1 = np.ascontiguousarray(I0), np.ascontiguousarray(I1)
    if I0.dtype != np.uint8 or I1.dtype != np.uint8 or I0.shape != I1.shape:
        raise TypeError("I0/I1 must be uint8 images of one shape")
    if not (I0.ndim == 2 or (I0.ndim == 3 and I0.shape[0] == 3)):
        raise TypeError("images must be (height, width) or (3, height, width)")
    return I0, I1, (1 if I0.ndim == 2 else 3)


def pyramidal_sgm(I0, I1, numPyd=5, *, device=0, out=None, **overrides):
    """[mvCurLevel, mvPyd, minC] = pyramidal_sgm(I0, I1, numPyd) (pyramidal_sgm.m:1).  Keyword overrides
    name the function's hard-coded parameters (P1, P2, aggHalfWinSize, verSearchHalfWinSize,
    horSearchHalfWinSize, enableDiagonal, totalPass, adaptiveP2; pyramidal_sgm.m:15-22).
    out: a previous call's result tuple to write into instead of allocating (the results of that call are overwritten): a loop
    over frames then hands the library pages that are already resident, as a MEX gateway's zero-filled outputs are."""
    lib = _lib.load()
    _bind(lib)
    I0, I1, ch = _check_images(I0, I1)
    H, W = I0.shape[-2:]
    prm = _params(lib, numPyd, device, overrides)
    sizes = [(W, H)]
    for _ in range(1, prm.numPyd):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    if out is not None:
        mv, mvPyd, minC = out
        ok = (mv.shape == (2, H, W) and mv.dtype == np.float64 and minC.shape == (H, W) and minC.dtype == np.uint32 and len(mvPyd) == len(sizes)
              and all(a.shape == (2, h, w) and a.dtype == np.float64 and a.flags.c_contiguous for a, (w, h) in zip(mvPyd, sizes)))
        if not ok or not mv.flags.c_contiguous or not minC.flags.c_contiguous:
            raise ValueError("out does not match this call's shapes")
    else:
        mv = np.empty((2, H, W), np.float64)                 # every element is written by the call
        minC = np.empty((H, W), np.uint32)
        mvPyd = [np.empty((2, h, w), np.float64) for (w, h) in sizes]
    ptrs = (C.c_void_p * len(mvPyd))(*[a.ctypes.data for a in mvPyd])
    check(lib.fsgm_pyramidal_sgm_host(ptr(I0), ptr(I1), W, H, ch, C.byref(prm), ptr(mv), ptr(minC), ptrs))
    return mv, mvPyd, minC


class PyramidPair(C.Structure):
    _fields_ = [("I0", C.c_void_p), ("I1", C.c_void_p), ("mv", C.c_void_p), ("minC", C.c_void_p), ("mvPyd", C.c_void_p)]


def _pairs_over_devices(fn, prm, pairs, devices):
    """pairs of one shape through fsgm_pyramidal_sgm(_ng)_batch_devices_host: pair i on devices[i % len(devices)]."""
    n = len(pairs)
    arr, keep, res = (PyramidPair * n)(), [], []
    shape = None
    for i, (I0, I1) in enumerate(pairs):
        I0, I1, ch = _check_images(I0, I1)
        H, W = I0.shape[-2:]
        if shape not in (None, (W, H, ch)):
            raise ValueError("all pairs of a batch must share one shape")
        shape = (W, H, ch)
        sizes = [(W, H)]
        for _ in range(1, prm.numPyd):
            sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
        mv, minC = np.zeros((2, H, W), np.float64), np.zeros((H, W), np.uint32)
        lv = [np.zeros((2, h, w), np.float64) for (w, h) in sizes]
        ptrs = (C.c_void_p * len(lv))(*[a.ctypes.data for a in lv])
        arr[i].I0, arr[i].I1, arr[i].mv, arr[i].minC = ptr(I0), ptr(I1), ptr(mv), ptr(minC)
        arr[i].mvPyd = C.cast(ptrs, C.c_void_p)
        keep.append((I0, I1, ptrs))
        res.append((mv, lv, minC))
    nd, darr = _lib.device_array(devices)
    W, H, ch = shape
    check(fn(n, arr, W, H, ch, C.byref(prm), nd, darr))
    return res


def pyramidal_sgm_batch(pairs, numPyd=5, *, devices=(0,), **overrides):
    """pyramidal_sgm for a list of (I0, I1) pairs of one shape, pair i on devices[i % len(devices)] (one host thread per
    entry inside the library, no collective); returns a list of (mvCurLevel, mvPyd, minC) like pyramidal_sgm."""
    lib = _lib.load()
    _bind(lib)
    lib.fsgm_pyramidal_sgm_batch_devices_host.argtypes = [C.c_int32, C.POINTER(PyramidPair), C.c_int32, C.c_int32, C.c_int32,
                                                          C.POINTER(PyramidParams), C.c_int32, C.POINTER(C.c_int32)]
    return _pairs_over_devices(lib.fsgm_pyramidal_sgm_batch_devices_host, _params(lib, numPyd, 0, overrides), pairs, devices)


class PyramidPlan:
    """Device-resident pyramid for one image shape: upload a pair, run, download any level."""

    def __init__(self, width, height, channels=1, numPyd=5, *, device=0, batch=1, **overrides):
        self.lib = _lib.load()
        _bind(self.lib)
        self.prm = _params(self.lib, numPyd, device, overrides)
        self.W, self.H, self.channels, self.batch = int(width), int(height), int(channels), int(batch)
        self._h = C.c_void_p()
        check(self.lib.fsgm_pyramid_plan_create_batch(C.byref(self._h), self.W, self.H, self.channels, C.byref(self.prm), self.batch))

    def close(self):
        if self._h:
            self.lib.fsgm_pyramid_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        check(self.lib.fsgm_pyramid_plan_level_size(self._h, int(level), C.byref(w), C.byref(h)))
        return w.value, h.value

    def upload(self, I0, I1, frame=0):
        I0, I1, ch = _check_images(I0, I1)
        if ch != self.channels or I0.shape[-2:] != (self.H, self.W):
            raise ValueError("shape mismatch with the plan")
        check(self.lib.fsgm_pyramid_plan_upload_frame(self._h, int(frame), ptr(I0), ptr(I1)))

    def run(self):
        check(self.lib.fsgm_pyramid_plan_run(self._h))

    def sync(self):
        """Wait for the queued work; plans started with run() before any sync() overlap on the device."""
        check(self.lib.fsgm_pyramid_plan_sync(self._h))

    def run_images(self):
        """Only impyramid / rgb2gray (the level images), no matching."""
        check(self.lib.fsgm_pyramid_plan_run_images(self._h))

    def download(self, level=1, frame=0):
        w, h = self.level_size(level)
        mv = np.empty((2, h, w), np.float64)
        minC = np.empty((h, w), np.uint32)
        check(self.lib.fsgm_pyramid_plan_download_frame(self._h, int(frame), int(level), ptr(mv), ptr(minC)))
        return mv, minC

    def download_gray(self, level=1, frame=0):
        """The gray image pair calc_pyd_cost_sgm saw at `level` (after impyramid / rgb2gray), of pair `frame` of the batch."""
        w, h = self.level_size(level)
        g0, g1 = np.empty((h, w), np.uint8), np.empty((h, w), np.uint8)
        check(self.lib.fsgm_pyramid_plan_download_gray_frame(self._h, int(frame), int(level), ptr(g0), ptr(g1)))
        return g0, g1

    def time(self, warmup=1, iters=5):
        ms = C.c_float()
        check(self.lib.fsgm_pyramid_plan_time(self._h, int(warmup), int(iters), C.byref(ms)))
        return float(ms.value)


class NgPyramidParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("numPyd", "P1", "P2", "halfSearchWinSize", "aggSize", "subPixelRefine", "device")]


def _bind_ng(lib):
    if getattr(lib, "_ng_pyramid_bound", False):
        return
    vp, i32 = C.c_void_p, C.c_int32
    lib.fsgm_ng_pyramid_params_default.restype = NgPyramidParams
    lib.fsgm_ng_pyramid_plan_create.argtypes = [C.POINTER(vp), i32, i32, i32, C.POINTER(NgPyramidParams)]
    lib.fsgm_ng_pyramid_plan_create_batch.argtypes = [C.POINTER(vp), i32, i32, i32, i32, C.POINTER(NgPyramidParams)]
    lib.fsgm_ng_pyramid_plan_upload_frame.argtypes = [vp, i32, vp, vp]
    lib.fsgm_ng_pyramid_plan_download_frame.argtypes = [vp, i32, i32, vp, vp]
    lib.fsgm_ng_pyramid_plan_destroy.argtypes = [vp]
    lib.fsgm_ng_pyramid_plan_destroy.restype = None
    lib.fsgm_ng_pyramid_plan_level_size.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.fsgm_ng_pyramid_plan_upload.argtypes = [vp, vp, vp]
    lib.fsgm_ng_pyramid_plan_run.argtypes = [vp]
    lib.fsgm_ng_pyramid_plan_download.argtypes = [vp, i32, vp, vp]
    lib.fsgm_ng_pyramid_plan_time.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
    lib.fsgm_pyramidal_sgm_ng_host.argtypes = [vp, vp, i32, i32, i32, C.POINTER(NgPyramidParams), vp, vp, vp]
    lib._ng_pyramid_bound = True


class NgPyramidPlan:
    """Device-resident level loop around calc_pyd_cost_sgm_ng for one image shape (see pyramidal_sgm_ng); `batch` image
    pairs stay resident and go through every level together."""

    def __init__(self, width, height, channels=1, numPyd=3, *, device=0, batch=1, **overrides):
        self.lib = _lib.load()
        _bind_ng(self.lib)
        prm = self.lib.fsgm_ng_pyramid_params_default()
        prm.numPyd, prm.device = int(numPyd), int(device)
        for k, v in overrides.items():
            if not hasattr(prm, k):
                raise TypeError(f"unknown pyramidal_sgm_ng parameter {k!r}")
            setattr(prm, k, int(v))
        self.prm = prm
        self.W, self.H, self.channels = int(width), int(height), int(channels)
        self.batch = int(batch)
        self._h = C.c_void_p()
        check(self.lib.fsgm_ng_pyramid_plan_create_batch(C.byref(self._h), self.W, self.H, self.channels, self.batch, C.byref(prm)))

    def close(self):
        if self._h:
            self.lib.fsgm_ng_pyramid_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        check(self.lib.fsgm_ng_pyramid_plan_level_size(self._h, int(level), C.byref(w), C.byref(h)))
        return w.value, h.value

    def upload(self, I0, I1, frame=0):
        I0, I1, ch = _check_images(I0, I1)
        if ch != self.channels or I0.shape[-2:] != (self.H, self.W):
            raise ValueError("shape mismatch with the plan")
        check(self.lib.fsgm_ng_pyramid_plan_upload_frame(self._h, int(frame), ptr(I0), ptr(I1)))

    def run(self):
        check(self.lib.fsgm_ng_pyramid_plan_run(self._h))

    def download(self, level=1, frame=0):
        w, h = self.level_size(level)
        flow = np.empty((2, h, w), np.float64)
        minC = np.empty((h, w), np.uint32)
        check(self.lib.fsgm_ng_pyramid_plan_download_frame(self._h, int(frame), int(level), ptr(flow), ptr(minC)))
        return flow, minC

    def time(self, warmup=1, iters=5):
        ms = C.c_float()
        check(self.lib.fsgm_ng_pyramid_plan_time(self._h, int(warmup), int(iters), C.byref(ms)))
        return float(ms.value)


def pyramidal_sgm_ng_batch(pairs, numPyd=3, *, devices=(0,), **overrides):
    """pyramidal_sgm_ng for a list of (I0, I1) pairs of one shape, pair i on devices[i % len(devices)]; returns a list of
    (flow of level 1, [flow per level], minC of level 1) like pyramidal_sgm_ng."""
    lib = _lib.load()
    _bind_ng(lib)
    lib.fsgm_pyramidal_sgm_ng_batch_devices_host.argtypes = [C.c_int32, C.POINTER(PyramidPair), C.c_int32, C.c_int32, C.c_int32,
                                                             C.POINTER(NgPyramidParams), C.c_int32, C.POINTER(C.c_int32)]
    prm = lib.fsgm_ng_pyramid_params_default()
    prm.numPyd = int(numPyd)
    for k, v in overrides.items():
        if not hasattr(prm, k):
            raise TypeError(f"unknown pyramidal_sgm_ng parameter {k!r}")
        setattr(prm, k, int(v))
    return _pairs_over_devices(lib.fsgm_pyramidal_sgm_ng_batch_devices_host, prm, pairs, devices)


def pyramidal_sgm_ng(I0, I1, numPyd=3, *, device=0, out=None, **overrides):
    """The level loop of pyramidal_sgm.m (:24-76) with calc_pyd_cost_sgm_ng swapped in for calc_pyd_cost_sgm
    (BASELINE config 4): the neighbour-guided MEX takes the previous level's flow as its hint map and returns
    the flow itself (calc_pyd_cost_sgm_ng.cpp:458-480).  Defaults are the argument values of ng_sgm.m:20
    (halfSearchWinSize=1, aggSize=2, subPixelRefine=0, P1=6, P2=32); keyword overrides name them.  Level images
    by impyramid 'reduce' / rgb2gray, hints of a finer level = 2*imresize(flow, 2, 'nearest') (pyramidal_sgm.m:72);
    everything stays on the device between levels.

    Returns (flow of level 1, [flow per level, coarsest first], minC of level 1)."""
    lib = _lib.load()
    _bind_ng(lib)
    I0, I1, ch = _check_images(I0, I1)
    H, W = I0.shape[-2:]
    prm = lib.fsgm_ng_pyramid_params_default()
    prm.numPyd, prm.device = int(numPyd), int(device)
    for k, v in overrides.items():
        if not hasattr(prm, k):
            raise TypeError(f"unknown pyramidal_sgm_ng parameter {k!r}")
        setattr(prm, k, int(v))
    sizes = [(W, H)]
    for _ in range(1, prm.numPyd):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    if out is not None:                                      # a previous call's result tuple, overwritten (see pyramidal_sgm)
        flow, lv, minC = out
        lv = list(lv)[::-1]                                  # returned coarsest first, filled finest first
        ok = (flow.shape == (2, H, W) and flow.dtype == np.float64 and minC.shape == (H, W) and minC.dtype == np.uint32 and len(lv) == len(sizes)
              and all(a.shape == (2, h, w) and a.dtype == np.float64 and a.flags.c_contiguous for a, (w, h) in zip(lv, sizes)))
        if not ok or not flow.flags.c_contiguous or not minC.flags.c_contiguous:
            raise ValueError("out does not match this call's shapes")
    else:
        flow = np.empty((2, H, W), np.float64)
        minC = np.empty((H, W), np.uint32)
        lv = [np.empty((2, h, w), np.float64) for (w, h) in sizes]
    ptrs = (C.c_void_p * len(lv))(*[a.ctypes.data for a in lv])
    check(lib.fsgm_pyramidal_sgm_ng_host(ptr(I0), ptr(I1), W, H, ch, C.byref(prm), ptr(flow), ptr(minC), ptrs))
    return flow, lv[::-1], minC
