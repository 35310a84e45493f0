"""Host-side mirror of the reference's calc_pyd_cost_sgm MEX interface
(calc_pyd_cost_sgm.cpp:439-510, called from pyramidal_sgm.m:50), on top of the C ABI.

Arrays follow the MEX's memory order: images (height, width) uint8 C-contiguous, preMv
(2, mvHeight, mvWidth) float64 with plane 0 = x, mvSub (2, height, width).
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import check, ptr


class PydIn(C.Structure):
    _fields_ = [("I1", C.c_void_p), ("I2", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("preMv", C.c_void_p), ("mvWidth", C.c_int32), ("mvHeight", C.c_int32),
                ("halfSearchWinSizeX", C.c_int32), ("halfSearchWinSizeY", C.c_int32), ("aggHalfWinSize", C.c_int32),
                ("subPixelRefine", C.c_int32), ("P1", C.c_int32), ("P2", C.c_int32),
                ("enableDiagnalPath", C.c_int32), ("totalPass", C.c_int32), ("adpativeP2", C.c_int32)]


class PydOut(C.Structure):
    _fields_ = [("bestD", C.c_void_p), ("minC", C.c_void_p), ("mvSub", C.c_void_p), ("C", C.c_void_p), ("S", C.c_void_p)]


def _bind(lib):
    if getattr(lib, "_pyd_bound", False):
        return
    vp, i32 = C.c_void_p, C.c_int32
    lib.fsgm_calc_pyd_cost_sgm_host.argtypes = [C.POINTER(PydIn), C.POINTER(PydOut), i32]
    lib.fsgm_calc_pyd_cost_sgm_batch_host.argtypes = [i32, C.POINTER(PydIn), C.POINTER(PydOut), i32]
    lib.fsgm_calc_pyd_cost_sgm_batch_devices_host.argtypes = [i32, C.POINTER(PydIn), C.POINTER(PydOut), i32, C.POINTER(i32)]
    lib.fsgm_pyd_plan_create.argtypes = [C.POINTER(vp)] + [i32] * 9
    lib.fsgm_pyd_plan_destroy.argtypes = [vp]
    lib.fsgm_pyd_plan_destroy.restype = None
    lib.fsgm_pyd_plan_set_params.argtypes = [vp] + [i32] * 6
    lib.fsgm_pyd_plan_upload.argtypes = [vp, i32, vp, vp, vp]
    lib.fsgm_pyd_plan_upload_cost.argtypes = [vp, i32, vp]
    lib.fsgm_pyd_plan_run.argtypes = [vp, i32]
    lib.fsgm_pyd_plan_download.argtypes = [vp, i32, vp, vp, vp]
    lib.fsgm_pyd_plan_download_cost.argtypes = [vp, i32, vp]
    lib.fsgm_pyd_plan_download_sum.argtypes = [vp, i32, vp]
    lib.fsgm_pyd_plan_time.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float)]
    lib._pyd_bound = True


def _check_inputs(I1, I2, preMv):
    I1, I2 = np.ascontiguousarray(I1), np.ascontiguousarray(I2)
    if I1.dtype != np.uint8 or I2.dtype != np.uint8 or I1.ndim != 2 or I1.shape != I2.shape:
        raise TypeError("I1/I2 must be uint8 images of one shape")
    preMv = np.asarray(preMv)
    if preMv.dtype != np.float64 or preMv.ndim != 3 or preMv.shape[0] != 2:
        raise TypeError("preMv must be float64 of shape (2, mvHeight, mvWidth)")
    H, W = I1.shape
    if preMv.shape[1] < H or preMv.shape[2] < W:
        raise ValueError("preMv must be at least as large as the image")
    return I1, I2, np.ascontiguousarray(preMv)


def calc_pyd_cost_sgm(I1, I2, preMv, halfSearchWinSizeX, halfSearchWinSizeY, aggHalfWinSize, subPixelRefine,
                      P1, P2, enableDiagnalPath, totalPass, adpativeP2, *, device=0, return_volumes=False):
    """[bestD, minC, mvSub] = calc_pyd_cost_sgm(I1, I2, preMv, halfSearchWinSizeX, halfSearchWinSizeY,
    aggHalfWinSize, subPixelRefine, P1, P2, enableDiagnalPath, totalPass, adpativeP2) -- the MEX's
    12 arguments in order."""
    lib = _lib.load()
    _bind(lib)
    I1, I2, preMv = _check_inputs(I1, I2, preMv)
    H, W = I1.shape
    for name, v in (("halfSearchWinSizeX", halfSearchWinSizeX), ("halfSearchWinSizeY", halfSearchWinSizeY),
                    ("aggHalfWinSize", aggHalfWinSize)):
        if int(v) != v or v < 0:
            raise ValueError(f"{name} must be a non-negative integer")
    D = (2 * int(halfSearchWinSizeX) + 1) * (2 * int(halfSearchWinSizeY) + 1)
    a = PydIn()
    a.I1, a.I2, a.width, a.height = ptr(I1), ptr(I2), W, H
    a.preMv, a.mvWidth, a.mvHeight = ptr(preMv), preMv.shape[2], preMv.shape[1]
    a.halfSearchWinSizeX, a.halfSearchWinSizeY, a.aggHalfWinSize = int(halfSearchWinSizeX), int(halfSearchWinSizeY), int(aggHalfWinSize)
    a.subPixelRefine, a.P1, a.P2 = int(subPixelRefine), int(P1), int(P2)
    a.enableDiagnalPath, a.totalPass, a.adpativeP2 = int(bool(enableDiagnalPath)), int(totalPass), int(bool(adpativeP2))
    bestD = np.zeros((H, W), np.uint32)
    minC = np.zeros((H, W), np.uint32)
    mvSub = np.zeros((2, H, W), np.float64)
    Cv = np.zeros((H, W, D), np.uint8) if return_volumes else None
    S = np.zeros((H, W, D), np.uint32) if return_volumes else None
    o = PydOut()
    o.bestD, o.minC, o.mvSub, o.C, o.S = ptr(bestD), ptr(minC), ptr(mvSub), ptr(Cv), ptr(S)
    check(lib.fsgm_calc_pyd_cost_sgm_host(C.byref(a), C.byref(o), int(device)))
    return (bestD, minC, mvSub, Cv, S) if return_volumes else (bestD, minC, mvSub)


def calc_pyd_cost_sgm_batch(frames, halfSearchWinSizeX, halfSearchWinSizeY, aggHalfWinSize, subPixelRefine,
                            P1, P2, enableDiagnalPath, totalPass, adpativeP2, *, device=0, devices=None):
    """frames: list of (I1, I2, preMv) of one shape and one parameter set, processed together; returns a list of
    (bestD, minC, mvSub).  devices: a device list -- frame i runs on devices[i % len(devices)], one host thread per entry
    inside the library, no collective (fsgm_calc_pyd_cost_sgm_batch_devices_host)."""
    lib = _lib.load()
    _bind(lib)
    n = len(frames)
    if n == 0:
        return []
    ins, outs, keep, res = (PydIn * n)(), (PydOut * n)(), [], []
    for i, (I1, I2, preMv) in enumerate(frames):
        I1, I2, preMv = _check_inputs(I1, I2, preMv)
        H, W = I1.shape
        a = ins[i]
        a.I1, a.I2, a.width, a.height = ptr(I1), ptr(I2), W, H
        a.preMv, a.mvWidth, a.mvHeight = ptr(preMv), preMv.shape[2], preMv.shape[1]
        a.halfSearchWinSizeX, a.halfSearchWinSizeY, a.aggHalfWinSize = int(halfSearchWinSizeX), int(halfSearchWinSizeY), int(aggHalfWinSize)
        a.subPixelRefine, a.P1, a.P2 = int(subPixelRefine), int(P1), int(P2)
        a.enableDiagnalPath, a.totalPass, a.adpativeP2 = int(bool(enableDiagnalPath)), int(totalPass), int(bool(adpativeP2))
        bestD, minC, mvSub = np.zeros((H, W), np.uint32), np.zeros((H, W), np.uint32), np.zeros((2, H, W), np.float64)
        o = outs[i]
        o.bestD, o.minC, o.mvSub, o.C, o.S = ptr(bestD), ptr(minC), ptr(mvSub), None, None
        keep.append((I1, I2, preMv))
        res.append((bestD, minC, mvSub))
    if devices is not None:
        nd, darr = _lib.device_array(devices)
        check(lib.fsgm_calc_pyd_cost_sgm_batch_devices_host(n, ins, outs, nd, darr))
    else:
        check(lib.fsgm_calc_pyd_cost_sgm_batch_host(n, ins, outs, int(device)))
    return res


class PydPlan:
    """Device-resident plan for `batch` frames of one pyramid level."""

    def __init__(self, width, height, mvWidth, mvHeight, rX, rY, rAgg, batch=1, *, device=0):
        self.lib = _lib.load()
        _bind(self.lib)
        self.W, self.H, self.mvW, self.mvH = int(width), int(height), int(mvWidth), int(mvHeight)
        self.Sx, self.Sy = 2 * int(rX) + 1, 2 * int(rY) + 1
        self.D = self.Sx * self.Sy
        self.batch = int(batch)
        self._h = C.c_void_p()
        check(self.lib.fsgm_pyd_plan_create(C.byref(self._h), self.W, self.H, self.mvW, self.mvH, int(rX), int(rY),
                                            int(rAgg), self.batch, int(device)))

    def close(self):
        if self._h:
            self.lib.fsgm_pyd_plan_destroy(self._h)
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

    def set_params(self, P1, P2, diagonal=1, totalPass=2, adaptiveP2=0, subpixel=0):
        check(self.lib.fsgm_pyd_plan_set_params(self._h, int(P1), int(P2), int(diagonal), int(totalPass),
                                                int(adaptiveP2), int(subpixel)))

    def upload(self, frame, I1, I2, preMv):
        I1, I2, preMv = _check_inputs(I1, I2, preMv)
        if I1.shape != (self.H, self.W) or preMv.shape != (2, self.mvH, self.mvW):
            raise ValueError("shape mismatch with the plan")
        check(self.lib.fsgm_pyd_plan_upload(self._h, frame, ptr(I1), ptr(I2), ptr(preMv)))

    def upload_cost(self, frame, Cv):
        Cv = np.ascontiguousarray(Cv)
        if Cv.dtype != np.uint8 or Cv.shape != (self.H, self.W, self.D):
            raise TypeError(f"C must be uint8 of shape {(self.H, self.W, self.D)}")
        check(self.lib.fsgm_pyd_plan_upload_cost(self._h, frame, ptr(Cv)))

    def run(self, stages=_lib.STAGE_ALL):
        check(self.lib.fsgm_pyd_plan_run(self._h, int(stages)))

    def download(self, frame):
        bestD = np.empty((self.H, self.W), np.uint32)
        minC = np.empty((self.H, self.W), np.uint32)
        mvSub = np.empty((2, self.H, self.W), np.float64)
        check(self.lib.fsgm_pyd_plan_download(self._h, frame, ptr(bestD), ptr(minC), ptr(mvSub)))
        return bestD, minC, mvSub

    def download_cost(self, frame):
        Cv = np.empty((self.H, self.W, self.D), np.uint8)
        check(self.lib.fsgm_pyd_plan_download_cost(self._h, frame, ptr(Cv)))
        return Cv

    def download_sum(self, frame):
        S = np.empty((self.H, self.W, self.D), np.uint32)
        check(self.lib.fsgm_pyd_plan_download_sum(self._h, frame, ptr(S)))
        return S

    def time(self, stages, warmup=1, iters=5):
        ms = C.c_float()
        check(self.lib.fsgm_pyd_plan_time(self._h, int(stages), int(warmup), int(iters), C.byref(ms)))
        return float(ms.value)
