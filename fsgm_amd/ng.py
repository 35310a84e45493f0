"""Host-side mirrors of the two neighbour-guided MEX interfaces on top of the C ABI:

  calc_pyd_cost_sgm_ng(I1, I2, preMv, halfSearchWinSize, aggSize, subPixelRefine, P1, P2)
      -> (minC, flow)            calc_pyd_cost_sgm_ng.cpp:448-523
  calc_cost_sgm_ng(I1, I2, preMv, halfSearchWinSize, aggSize, subPixelRefine, P1, P2)
      -> (minC, flow)            calc_cost_sgm_ng.cpp:484-526 (arguments 3-6 are ignored there too)

Arrays follow the MEX's memory order (see fsgm_amd.epi).
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import check, ptr


class NgIn(C.Structure):
    _fields_ = [("I1", C.c_void_p), ("I2", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("preMv", C.c_void_p), ("mvWidth", C.c_int32), ("mvHeight", C.c_int32),
                ("halfSearchWinSize", C.c_int32), ("aggSize", C.c_int32), ("subPixelRefine", C.c_int32),
                ("P1", C.c_int32), ("P2", C.c_int32)]


class NgOut(C.Structure):
    _fields_ = [("minC", C.c_void_p), ("flow", C.c_void_p), ("S", C.c_void_p)]


class OtfIn(C.Structure):
    _fields_ = [("I1", C.c_void_p), ("I2", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("P1", C.c_int32), ("P2", C.c_int32), ("rand_stream", C.c_void_p)]


class OtfOut(C.Structure):
    _fields_ = [("minC", C.c_void_p), ("flow", C.c_void_p)]


def _bind(lib):
    if getattr(lib, "_ng_bound", False):
        return
    lib.fsgm_calc_pyd_cost_sgm_ng_host.argtypes = [C.POINTER(NgIn), C.POINTER(NgOut), C.c_int32]
    lib.fsgm_calc_cost_sgm_ng_host.argtypes = [C.POINTER(OtfIn), C.POINTER(OtfOut), C.c_int32]
    lib.fsgm_calc_pyd_cost_sgm_ng_batch_host.argtypes = [C.c_int32, C.POINTER(NgIn), C.POINTER(NgOut), C.c_int32]
    lib.fsgm_calc_cost_sgm_ng_batch_host.argtypes = [C.c_int32, C.POINTER(OtfIn), C.POINTER(OtfOut), C.c_int32]
    lib.fsgm_calc_pyd_cost_sgm_ng_batch_devices_host.argtypes = [C.c_int32, C.POINTER(NgIn), C.POINTER(NgOut), C.c_int32, C.POINTER(C.c_int32)]
    lib.fsgm_calc_cost_sgm_ng_batch_devices_host.argtypes = [C.c_int32, C.POINTER(OtfIn), C.POINTER(OtfOut), C.c_int32, C.POINTER(C.c_int32)]
    lib.fsgm_sgm_ng_rand_draws.argtypes = [C.c_int32, C.c_int32]
    lib.fsgm_sgm_ng_rand_draws.restype = C.c_int64
    lib._ng_bound = True


def _images(I1, I2):
    I1, I2 = np.ascontiguousarray(I1), np.ascontiguousarray(I2)
    if I1.dtype != np.uint8 or I2.dtype != np.uint8 or I1.ndim != 2 or I1.shape != I2.shape:
        raise TypeError("I1/I2 must be uint8 images of one shape")
    return I1, I2


def calc_pyd_cost_sgm_ng(I1, I2, preMv, halfSearchWinSize, aggSize, subPixelRefine, P1, P2, *, device=0,
                         return_sum=False):
    lib = _lib.load()
    _bind(lib)
    I1, I2 = _images(I1, I2)
    preMv = np.asarray(preMv)
    if preMv.dtype != np.float64 or preMv.ndim != 3 or preMv.shape[0] != 2:
        raise TypeError("preMv must be float64 of shape (2, mvHeight, mvWidth)")
    preMv = np.ascontiguousarray(preMv)
    H, W = I1.shape
    r = int(halfSearchWinSize)
    D = 9 * (2 * r + 1) ** 2
    a = NgIn()
    a.I1, a.I2, a.width, a.height = ptr(I1), ptr(I2), W, H
    a.preMv, a.mvWidth, a.mvHeight = ptr(preMv), preMv.shape[2], preMv.shape[1]
    a.halfSearchWinSize, a.aggSize, a.subPixelRefine, a.P1, a.P2 = r, int(aggSize), int(subPixelRefine), int(P1), int(P2)
    minC = np.zeros((H, W), np.uint32)
    flow = np.zeros((2, H, W), np.float64)
    S = np.zeros((H, W, D), np.uint32) if return_sum else None
    o = NgOut()
    o.minC, o.flow, o.S = ptr(minC), ptr(flow), ptr(S)
    check(lib.fsgm_calc_pyd_cost_sgm_ng_host(C.byref(a), C.byref(o), int(device)))
    return (minC, flow, S) if return_sum else (minC, flow)


def sgm_ng_rand_draws(width, height):
    lib = _lib.load()
    _bind(lib)
    return int(lib.fsgm_sgm_ng_rand_draws(int(width), int(height)))


def calc_cost_sgm_ng(I1, I2, preMv=None, halfSearchWinSize=1, aggSize=2, subPixelRefine=0, P1=6, P2=32, *,
                     rand_stream=None, device=0):
    """rand_stream: the libc rand() values the reference would draw (8 per pixel, raster order);
    None = the library draws them from libc rand() itself, like the reference."""
    lib = _lib.load()
    _bind(lib)
    I1, I2 = _images(I1, I2)
    H, W = I1.shape
    rs = None
    if rand_stream is not None:
        rs = np.ascontiguousarray(rand_stream, np.int32)
        if rs.size < sgm_ng_rand_draws(W, H):
            raise ValueError("rand_stream too short")
    a = OtfIn()
    a.I1, a.I2, a.width, a.height, a.P1, a.P2, a.rand_stream = ptr(I1), ptr(I2), W, H, int(P1), int(P2), ptr(rs)
    minC = np.zeros((H, W), np.uint32)
    flow = np.zeros((2, H, W), np.float64)
    o = OtfOut()
    o.minC, o.flow = ptr(minC), ptr(flow)
    check(lib.fsgm_calc_cost_sgm_ng_host(C.byref(a), C.byref(o), int(device)))
    return minC, flow


def calc_pyd_cost_sgm_ng_batch(frames, halfSearchWinSize, aggSize, subPixelRefine, P1, P2, *, device=0, devices=None):
    """frames: list of (I1, I2, preMv) of one shape; one launch sequence for all of them.
    Returns a list of (minC, flow).  devices: a device list (frame i on devices[i % len], fsgm_amd.calc_cost_sgm_batch)."""
    lib = _lib.load()
    _bind(lib)
    n = len(frames)
    ins, outs, keep, res = (NgIn * n)(), (NgOut * n)(), [], []
    for i, (I1, I2, preMv) in enumerate(frames):
        I1, I2 = _images(I1, I2)
        preMv = np.ascontiguousarray(preMv, np.float64)
        if preMv.ndim != 3 or preMv.shape[0] != 2:
            raise TypeError("preMv must be float64 of shape (2, mvHeight, mvWidth)")
        H, W = I1.shape
        a = ins[i]
        a.I1, a.I2, a.width, a.height = ptr(I1), ptr(I2), W, H
        a.preMv, a.mvWidth, a.mvHeight = ptr(preMv), preMv.shape[2], preMv.shape[1]
        a.halfSearchWinSize, a.aggSize, a.subPixelRefine, a.P1, a.P2 = int(halfSearchWinSize), int(aggSize), int(subPixelRefine), int(P1), int(P2)
        minC, flow = np.zeros((H, W), np.uint32), np.zeros((2, H, W), np.float64)
        outs[i].minC, outs[i].flow, outs[i].S = ptr(minC), ptr(flow), None
        keep.append((I1, I2, preMv))
        res.append((minC, flow))
    if devices is not None:
        nd, darr = _lib.device_array(devices)
        check(lib.fsgm_calc_pyd_cost_sgm_ng_batch_devices_host(n, ins, outs, nd, darr))
    else:
        check(lib.fsgm_calc_pyd_cost_sgm_ng_batch_host(n, ins, outs, int(device)))
    return res


def calc_cost_sgm_ng_batch(frames, P1=6, P2=32, *, device=0, devices=None):
    """frames: list of (I1, I2, rand_stream) of one shape (rand_stream as for calc_cost_sgm_ng, not None:
    frames of a batch run side by side, so there is no 'order of draws' between them).  Returns a list of (minC, flow)."""
    lib = _lib.load()
    _bind(lib)
    n = len(frames)
    ins, outs, keep, res = (OtfIn * n)(), (OtfOut * n)(), [], []
    for i, (I1, I2, rand_stream) in enumerate(frames):
        I1, I2 = _images(I1, I2)
        H, W = I1.shape
        rs = np.ascontiguousarray(rand_stream, np.int32)
        if rs.size < sgm_ng_rand_draws(W, H):
            raise ValueError("rand_stream too short")
        a = ins[i]
        a.I1, a.I2, a.width, a.height, a.P1, a.P2, a.rand_stream = ptr(I1), ptr(I2), W, H, int(P1), int(P2), ptr(rs)
        minC, flow = np.zeros((H, W), np.uint32), np.zeros((2, H, W), np.float64)
        outs[i].minC, outs[i].flow = ptr(minC), ptr(flow)
        keep.append((I1, I2, rs))
        res.append((minC, flow))
    if devices is not None:
        nd, darr = _lib.device_array(devices)
        check(lib.fsgm_calc_cost_sgm_ng_batch_devices_host(n, ins, outs, nd, darr))
    else:
        check(lib.fsgm_calc_cost_sgm_ng_batch_host(n, ins, outs, int(device)))
    return res
