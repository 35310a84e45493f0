"""Host-side mirror of the reference's post-processing functions -- speckle_filter.m,
calc_disp_from_first.m, forward_backward_check.m, scanline_in_fill.m, vzInd2Disp.m, chained by
test.m:45-50 -- on top of the C ABI.  Same names and argument order as the MATLAB functions.

Maps are (height, width) float64 C-contiguous with NaN = invalid; Pd0 / normDirect are
(2, height, width) with plane 0 = x, Pd0 in MATLAB's 1-based pixel coordinates.
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import check, ptr


def _bind(lib):
    if getattr(lib, "_post_bound", False):
        return
    vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double
    lib.fsgm_speckle_filter_host.argtypes = [vp, i32, i32, f64, f64, vp, vp, i32]
    lib.fsgm_calc_disp_from_first_host.argtypes = [vp, i32, i32, vp, vp, vp, f64, f64, vp, i32]
    lib.fsgm_forward_backward_check_host.argtypes = [vp, vp, i32, i32, vp, vp, vp, f64, f64, vp, i32]
    lib.fsgm_scanline_in_fill_host.argtypes = [vp, i32, i32, vp, i32]
    lib.fsgm_vzind2disp_host.argtypes = [vp, vp, i32, i32, f64, f64, vp, i32]
    lib.fsgm_vmf_host.argtypes = [vp, i32, i32, i32, vp, i32]
    lib.fsgm_epi_postprocess_host.argtypes = [vp, i32, i32, vp, vp, vp, f64, f64, f64, vp, vp, vp, i32]
    lib.fsgm_post_plan_create.argtypes = [C.POINTER(vp), i32, i32, i32]
    lib.fsgm_post_plan_destroy.argtypes = [vp]
    lib.fsgm_post_plan_destroy.restype = None
    lib.fsgm_post_plan_upload.argtypes = [vp, vp, vp, vp, vp]
    lib.fsgm_post_plan_run.argtypes = [vp, f64, f64, f64]
    lib.fsgm_post_plan_download.argtypes = [vp, vp, vp, vp]
    lib.fsgm_post_plan_time.argtypes = [vp, f64, f64, f64, i32, i32, C.POINTER(C.c_float)]
    lib._post_bound = True


def _lib_bound():
    lib = _lib.load()
    _bind(lib)
    return lib


def _map(a, name, shape=None):
    a = np.ascontiguousarray(a)
    if a.dtype != np.float64 or a.ndim != 2:
        raise TypeError(f"{name} must be a float64 (height, width) map")
    if shape is not None and a.shape != shape:
        raise ValueError(f"{name} must have shape {shape}")
    return a


def _geom(Pd0, normDirect, O, shape):
    Pd0, normDirect = np.ascontiguousarray(Pd0), np.ascontiguousarray(normDirect)
    for a, name in ((Pd0, "Pd0"), (normDirect, "normDirect")):
        if a.dtype != np.float64 or a.shape != (2,) + shape:
            raise TypeError(f"{name} must be float64 of shape {(2,) + shape}")
    return Pd0, normDirect, _map(O, "O", shape)


def speckle_filter(image, maxDiff=2, maxSpeckleSize=100, *, device=0):
    """[imageFiltered, labelImage] = speckle_filter(image, maxDiff, maxSpeckleSize)  (speckle_filter.m:1)"""
    lib = _lib_bound()
    image = _map(image, "image")
    H, W = image.shape
    out = np.empty_like(image)
    labels = np.empty((H, W), np.int32)
    check(lib.fsgm_speckle_filter_host(ptr(image), W, H, float(maxDiff), float(maxSpeckleSize), ptr(out), ptr(labels), int(device)))
    return out, labels


def calc_disp_from_first(D1, Pd0, normDirect, O, vMax, n, *, device=0):
    """D2 = calc_disp_from_first(D1, Pd0, normDirect, O, vMax, n)  (calc_disp_from_first.m:1)"""
    lib = _lib_bound()
    D1 = _map(D1, "D1")
    H, W = D1.shape
    Pd0, normDirect, O = _geom(Pd0, normDirect, O, D1.shape)
    D2 = np.empty_like(D1)
    check(lib.fsgm_calc_disp_from_first_host(ptr(D1), W, H, ptr(Pd0), ptr(normDirect), ptr(O), float(vMax), float(n), ptr(D2), int(device)))
    return D2


def forward_backward_check(D1, D2, Pd0, normDirect, O, vMax, n, *, device=0):
    """D1 = forward_backward_check(D1, D2, Pd0, normDirect, O, vMax, n)  (forward_backward_check.m:1)"""
    lib = _lib_bound()
    D1 = _map(D1, "D1")
    H, W = D1.shape
    D2 = _map(D2, "D2", D1.shape)
    Pd0, normDirect, O = _geom(Pd0, normDirect, O, D1.shape)
    out = np.empty_like(D1)
    check(lib.fsgm_forward_backward_check_host(ptr(D1), ptr(D2), W, H, ptr(Pd0), ptr(normDirect), ptr(O), float(vMax), float(n),
                                               ptr(out), int(device)))
    return out


def scanline_in_fill(input, *, device=0):
    """output = scanline_in_fill(input)  (scanline_in_fill.m:2), one channel"""
    lib = _lib_bound()
    a = _map(input, "input")
    H, W = a.shape
    out = np.empty_like(a)
    check(lib.fsgm_scanline_in_fill_host(ptr(a), W, H, ptr(out), int(device)))
    return out


def vzInd2Disp(w, O, vMax, n, *, device=0):
    """D = vzInd2Disp(w, O, vMax, n)  (vzInd2Disp.m:1)"""
    lib = _lib_bound()
    w = _map(w, "w")
    H, W = w.shape
    O = _map(O, "O", w.shape)
    D = np.empty_like(w)
    check(lib.fsgm_vzind2disp_host(ptr(w), ptr(O), W, H, float(vMax), float(n), ptr(D), int(device)))
    return D


def vmf(flow, *, device=0):
    """flowMed = vmf(flow)  (vmf.m:1): 5x5 median per channel; flow (channels, height, width) float64"""
    lib = _lib_bound()
    flow = np.ascontiguousarray(flow)
    if flow.dtype != np.float64 or flow.ndim != 3 or not 1 <= flow.shape[0] <= 3:
        raise TypeError("flow must be float64 of shape (1..3, height, width)")
    ch, H, W = flow.shape
    out = np.empty_like(flow)
    check(lib.fsgm_vmf_host(ptr(flow), W, H, ch, ptr(out), int(device)))
    return out


def epi_postprocess(D1, Pd0, normDirect, O, vMax, n, dMax, *, device=0):
    """test.m:45-50 in one device-resident call.  Returns (filterD1, filterD2, filterdisparites)."""
    lib = _lib_bound()
    D1 = _map(D1, "D1")
    H, W = D1.shape
    Pd0, normDirect, O = _geom(Pd0, normDirect, O, D1.shape)
    f1, f2, disp = np.empty_like(D1), np.empty_like(D1), np.empty_like(D1)
    check(lib.fsgm_epi_postprocess_host(ptr(D1), W, H, ptr(Pd0), ptr(normDirect), ptr(O), float(vMax), float(n), float(dMax),
                                        ptr(f1), ptr(f2), ptr(disp), int(device)))
    return f1, f2, disp


class PostPlan:
    """Device-resident post-processing chain for one map shape."""

    def __init__(self, width, height, *, device=0):
        self.lib = _lib_bound()
        self.W, self.H = int(width), int(height)
        self._h = C.c_void_p()
        check(self.lib.fsgm_post_plan_create(C.byref(self._h), self.W, self.H, int(device)))

    def close(self):
        if self._h:
            self.lib.fsgm_post_plan_destroy(self._h)
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

    def upload(self, D1=None, Pd0=None, normDirect=None, O=None):
        shape = (self.H, self.W)
        D1 = None if D1 is None else _map(D1, "D1", shape)
        O = None if O is None else _map(O, "O", shape)
        Pd0 = None if Pd0 is None else np.ascontiguousarray(Pd0, np.float64)
        normDirect = None if normDirect is None else np.ascontiguousarray(normDirect, np.float64)
        check(self.lib.fsgm_post_plan_upload(self._h, ptr(D1), ptr(Pd0), ptr(normDirect), ptr(O)))

    def run(self, vMax, n, dMax):
        check(self.lib.fsgm_post_plan_run(self._h, float(vMax), float(n), float(dMax)))

    def download(self):
        f1, f2, disp = (np.empty((self.H, self.W), np.float64) for _ in range(3))
        check(self.lib.fsgm_post_plan_download(self._h, ptr(f1), ptr(f2), ptr(disp)))
        return f1, f2, disp

    def time(self, vMax, n, dMax, warmup=1, iters=5):
        ms = C.c_float()
        check(self.lib.fsgm_post_plan_time(self._h, float(vMax), float(n), float(dMax), int(warmup), int(iters), C.byref(ms)))
        return float(ms.value)
