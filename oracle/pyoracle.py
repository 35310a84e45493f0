"""ctypes binding of the CPU oracle (oracle/libfsgm_oracle.so) and, when present, of the
reference's own census built into oracle/_ref/.

TEST INFRASTRUCTURE: imported only from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product (fsgm_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfsgm_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libfsgm_ref_common.so")
_lib = None
_ref = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def ref_census_available():
    return os.path.exists(_REF)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def census(img, half=2):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.zeros((H, W), np.uint32)
    lib().fsgm_oracle_census(_p(img), _p(out), W, H, half)
    return out


def ref_census(img, half=2):
    """The reference's own census() (common.cpp:3), compiled unmodified into oracle/_ref/."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF)
    img = np.ascontiguousarray(img, np.uint8).copy()
    H, W = img.shape
    out = np.zeros((H, W), np.uint32)
    fn = getattr(_ref, "_Z6censusPhPjiii")        # void census(PixelType*, unsigned*, int, int, int)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    fn.restype = None
    fn(_p(img), _p(out), W, H, half)
    return out


def epi_cost(I1, I2, D, vMax, pd0, nd, off, want_raw=False):
    H, W = I1.shape
    Cv = np.zeros((H, W, D), np.uint8)
    raw = np.zeros((H, W, D), np.uint8) if want_raw else None
    lib().fsgm_oracle_epi_cost(_p(Cv), _p(raw), _p(np.ascontiguousarray(I1)), _p(np.ascontiguousarray(I2)),
                               W, H, D, C.c_double(vMax), _p(np.ascontiguousarray(pd0)),
                               _p(np.ascontiguousarray(nd)), _p(np.ascontiguousarray(off)))
    return (Cv, raw) if want_raw else Cv


def epi_aggregate(Cv, P1, P2, paths):
    Cv = np.ascontiguousarray(Cv, np.uint8)
    H, W, D = Cv.shape
    S = np.zeros(H * W * D + 1, np.uint32)
    lib().fsgm_oracle_epi_aggregate(_p(S), _p(Cv), W, H, D, int(P1), int(P2), int(paths))
    return S


def epi_wta(S, W, H, D, subpixel=1):
    bestD = np.zeros((H, W), np.uint32)
    minC = np.zeros((H, W), np.uint32)
    lib().fsgm_oracle_epi_wta(_p(bestD), _p(minC), _p(S), W, H, D, int(subpixel))
    return bestD, minC


def epi_vz_to_disp(bestD, off, vMax, n):
    out = np.ascontiguousarray(bestD, np.uint32).copy()
    H, W = out.shape
    lib().fsgm_oracle_epi_vz_to_disp(_p(out), W, H, _p(np.ascontiguousarray(off)), C.c_double(vMax), int(n))
    return out


def calc_cost_sgm(I1, I2, D, vMax, pd0, nd, off, P1, P2, paths=4, want_volumes=False):
    I1 = np.ascontiguousarray(I1, np.uint8)
    I2 = np.ascontiguousarray(I2, np.uint8)
    H, W = I1.shape
    bestD = np.zeros((H, W), np.uint32)
    minC = np.zeros((H, W), np.uint32)
    Cv = np.zeros((H, W, D), np.uint8) if want_volumes else None
    S = np.zeros(H * W * D + 1, np.uint32) if want_volumes else None
    lib().fsgm_oracle_calc_cost_sgm(_p(bestD), _p(minC), _p(I1), _p(I2), W, H, int(D), C.c_double(vMax),
                                    _p(np.ascontiguousarray(pd0)), _p(np.ascontiguousarray(nd)),
                                    _p(np.ascontiguousarray(off)), int(P1), int(P2), int(paths), _p(Cv), _p(S))
    if want_volumes:
        return bestD, minC, Cv, S[:-1].reshape(H, W, D)
    return bestD, minC
