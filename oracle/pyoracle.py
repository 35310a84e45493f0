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


# ---------------------------------------------------------------- calc_pyd_cost_sgm
def pyd_cost(cen1, cen2, preMv, rAgg, rX, rY):
    H, W = cen1.shape
    mvH, mvW = preMv.shape[1:]
    D = (2 * rX + 1) * (2 * rY + 1)
    Cv = np.zeros((H, W, D), np.uint8)
    lib().fsgm_oracle_pyd_cost(_p(Cv), _p(np.ascontiguousarray(cen1)), _p(np.ascontiguousarray(cen2)), W, H,
                               _p(np.ascontiguousarray(preMv)), mvW, mvH, int(rAgg), int(rX), int(rY))
    return Cv


def pyd_aggregate(I1, Cv, preMv, Sx, Sy, P1, P2, diagonal=1, totalPass=2, adaptiveP2=0):
    H, W, D = Cv.shape
    mvH, mvW = preMv.shape[1:]
    S = np.zeros((H, W, D), np.uint32)
    lib().fsgm_oracle_pyd_aggregate(_p(S), _p(np.ascontiguousarray(I1)), _p(np.ascontiguousarray(Cv)), W, H,
                                    _p(np.ascontiguousarray(preMv)), mvW, mvH, int(Sx), int(Sy), int(P1), int(P2),
                                    int(diagonal), int(totalPass), int(adaptiveP2))
    return S


def pyd_wta(S, Sx, Sy, subpixel):
    H, W, D = S.shape
    bestD = np.zeros((H, W), np.uint32)
    minC = np.zeros((H, W), np.uint32)
    mvSub = np.zeros((2, H, W), np.float64)
    lib().fsgm_oracle_pyd_wta(_p(bestD), _p(minC), _p(mvSub), _p(np.ascontiguousarray(S)), W, H, int(Sx), int(Sy), int(subpixel))
    return bestD, minC, mvSub


def calc_pyd_cost_sgm(I1, I2, preMv, rX, rY, rAgg, subpixel, P1, P2, diagonal=1, totalPass=2, adaptiveP2=0,
                      want_volumes=False):
    I1 = np.ascontiguousarray(I1, np.uint8)
    I2 = np.ascontiguousarray(I2, np.uint8)
    preMv = np.ascontiguousarray(preMv, np.float64)
    H, W = I1.shape
    mvH, mvW = preMv.shape[1:]
    D = (2 * rX + 1) * (2 * rY + 1)
    bestD = np.zeros((H, W), np.uint32)
    minC = np.zeros((H, W), np.uint32)
    mvSub = np.zeros((2, H, W), np.float64)
    Cv = np.zeros((H, W, D), np.uint8) if want_volumes else None
    S = np.zeros((H, W, D), np.uint32) if want_volumes else None
    lib().fsgm_oracle_calc_pyd_cost_sgm(_p(bestD), _p(minC), _p(mvSub), _p(I1), _p(I2), W, H, _p(preMv), mvW, mvH,
                                        int(rX), int(rY), int(rAgg), int(subpixel), int(P1), int(P2),
                                        int(diagonal), int(totalPass), int(adaptiveP2), _p(Cv), _p(S))
    return (bestD, minC, mvSub, Cv, S) if want_volumes else (bestD, minC, mvSub)


# ---------------------------------------------------------------- pyramidal_sgm.m
def impyramid_reduce(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.zeros(((H + 1) // 2, (W + 1) // 2), np.uint8)
    lib().fsgm_oracle_impyramid_reduce(_p(out), _p(img), W, H)
    return out


def rgb2gray(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    _, H, W = rgb.shape
    out = np.zeros((H, W), np.uint8)
    lib().fsgm_oracle_rgb2gray(_p(out), _p(rgb), W, H)
    return out


def pyramid_sizes(W, H, numPyd):
    sizes = [(W, H)]
    for _ in range(1, numPyd):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    return sizes


def pyramidal_sgm(I0, I1, numPyd=5, P1=6, P2=32, aggHalfWinSize=2, ver=5, hor=5, diagonal=1, totalPass=2, adaptiveP2=0):
    """pyramidal_sgm.m: I0/I1 uint8 (H, W) or (3, H, W).  Returns (mv (2,H,W), minC (H,W), [mv per level])."""
    I0 = np.ascontiguousarray(I0, np.uint8)
    I1 = np.ascontiguousarray(I1, np.uint8)
    ch = 1 if I0.ndim == 2 else I0.shape[0]
    H, W = I0.shape[-2:]
    mv = np.zeros((2, H, W), np.float64)
    minC = np.zeros((H, W), np.uint32)
    lv = [np.zeros((2, h, w), np.float64) for (w, h) in pyramid_sizes(W, H, numPyd)]
    ptrs = (C.c_void_p * numPyd)(*[a.ctypes.data for a in lv])
    lib().fsgm_oracle_pyramidal_sgm(_p(mv), _p(minC), ptrs, _p(I0), _p(I1), W, H, ch, int(numPyd), int(P1), int(P2),
                                    int(aggHalfWinSize), int(ver), int(hor), int(diagonal), int(totalPass), int(adaptiveP2))
    return mv, minC, lv


# ---------------------------------------------------------------- epipolar driver, dense half
def epipolar_maps(F, Hm, epipole, direction, W, H):
    F, Hm = np.ascontiguousarray(F, np.float64), np.ascontiguousarray(Hm, np.float64)
    Pd0, nd, rflow = (np.zeros((2, H, W)) for _ in range(3))
    off = np.zeros((H, W))
    lib().fsgm_oracle_epipolar_maps(_p(Pd0), _p(nd), _p(off), _p(rflow), _p(F), _p(Hm), C.c_double(epipole[0]),
                                    C.c_double(epipole[1]), int(direction), W, H)
    return Pd0, nd, off, rflow


def epipolar_sgm_of(I0, I1, F, Hm, epipole, direction, dMax=64, vMax=0.3, paths=4):
    I0, I1 = np.ascontiguousarray(I0, np.uint8), np.ascontiguousarray(I1, np.uint8)
    ch = 1 if I0.ndim == 2 else 3
    H, W = I0.shape[-2:]
    F, Hm = np.ascontiguousarray(F, np.float64), np.ascontiguousarray(Hm, np.float64)
    flow = np.zeros((3, H, W))
    minC = np.zeros((H, W), np.uint32)
    lib().fsgm_oracle_epipolar_sgm_of(_p(flow), _p(minC), _p(I0), _p(I1), W, H, ch, _p(F), _p(Hm), C.c_double(epipole[0]),
                                      C.c_double(epipole[1]), int(direction), int(dMax), C.c_double(vMax), int(paths))
    return flow, minC


# ---------------------------------------------------------------- post-processing (test.m:45-50)
def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def vzind2disp(w, O, vMax, n):
    w, O = _f64(w), _f64(O)
    D = np.zeros_like(w)
    lib().fsgm_oracle_vzind2disp(_p(D), _p(w), _p(O), int(w.size), C.c_double(vMax), C.c_double(n))
    return D


def speckle_filter(image, maxDiff=2, maxSpeckleSize=100):
    image = _f64(image)
    H, W = image.shape
    out = np.zeros_like(image)
    labels = np.zeros((H, W), np.int32)
    lib().fsgm_oracle_speckle_filter(_p(out), _p(labels), _p(image), W, H, C.c_double(maxDiff), C.c_double(maxSpeckleSize))
    return out, labels


def calc_disp_from_first(D1, Pd0, nd, O, vMax, n):
    D1 = _f64(D1)
    H, W = D1.shape
    D2 = np.zeros_like(D1)
    lib().fsgm_oracle_calc_disp_from_first(_p(D2), _p(D1), W, H, _p(_f64(Pd0)), _p(_f64(nd)), _p(_f64(O)), C.c_double(vMax), C.c_double(n))
    return D2


def forward_backward_check(D1, D2, Pd0, nd, O, vMax, n):
    D1 = _f64(D1)
    H, W = D1.shape
    out = np.zeros_like(D1)
    lib().fsgm_oracle_forward_backward_check(_p(out), _p(D1), _p(_f64(D2)), W, H, _p(_f64(Pd0)), _p(_f64(nd)), _p(_f64(O)),
                                             C.c_double(vMax), C.c_double(n))
    return out


def scanline_in_fill(a):
    a = _f64(a)
    H, W = a.shape
    out = np.zeros_like(a)
    lib().fsgm_oracle_scanline_in_fill(_p(out), _p(a), W, H)
    return out


def vmf(flow):
    flow = _f64(flow)
    ch, H, W = flow.shape
    out = np.zeros_like(flow)
    lib().fsgm_oracle_vmf(_p(out), _p(flow), W, H, ch)
    return out


def postprocess(D1, Pd0, nd, O, vMax, n, dMax):
    D1 = _f64(D1)
    H, W = D1.shape
    f1, f2, disp = np.zeros_like(D1), np.zeros_like(D1), np.zeros_like(D1)
    lib().fsgm_oracle_postprocess(_p(f1), _p(f2), _p(disp), _p(D1), W, H, _p(_f64(Pd0)), _p(_f64(nd)), _p(_f64(O)),
                                  C.c_double(vMax), C.c_double(n), C.c_double(dMax))
    return f1, f2, disp


# ---------------------------------------------------------------- calc_pyd_cost_sgm_ng
CAND = np.dtype([("mvx", np.int32), ("mvy", np.int32), ("cost", np.int32)])


def calc_pyd_cost_sgm_ng(I1, I2, preMv, halfSearchWinSize, aggSize, subpixel, P1, P2, want_volumes=False):
    I1 = np.ascontiguousarray(I1, np.uint8)
    I2 = np.ascontiguousarray(I2, np.uint8)
    preMv = np.ascontiguousarray(preMv, np.float64)
    H, W = I1.shape
    mvH, mvW = preMv.shape[1:]
    r = int(halfSearchWinSize)
    D = 9 * (2 * r + 1) ** 2
    minC = np.zeros((H, W), np.uint32)
    flow = np.zeros((2, H, W), np.float64)
    Cc = np.zeros((H, W, D), CAND) if want_volumes else None
    S = np.zeros((H, W, D), np.uint32) if want_volumes else None
    lib().fsgm_oracle_calc_pyd_cost_sgm_ng(_p(minC), _p(flow), _p(I1), _p(I2), W, H, _p(preMv), mvW, mvH,
                                           C.c_double(halfSearchWinSize), C.c_double(aggSize), int(subpixel),
                                           int(P1), int(P2), _p(Cc), _p(S))
    return (minC, flow, Cc, S) if want_volumes else (minC, flow)


# ---------------------------------------------------------------- calc_cost_sgm_ng
def glibc_rand_stream(n, seed=1):
    """n values of libc rand() after srand(seed) -- what the reference draws on this platform.
    (Restores nothing: rand() state is process-global, exactly as in a MATLAB session.)"""
    libc = C.CDLL(None)
    libc.srand(C.c_uint(seed))
    libc.rand.restype = C.c_int
    return np.array([libc.rand() for _ in range(n)], np.int32)


def sgm_ng_rand_draws(W, H):
    lib().fsgm_oracle_sgm_ng_rand_draws.restype = C.c_int64
    return int(lib().fsgm_oracle_sgm_ng_rand_draws(W, H))


def calc_cost_sgm_ng(I1, I2, P1, P2, rand_stream):
    I1 = np.ascontiguousarray(I1, np.uint8)
    I2 = np.ascontiguousarray(I2, np.uint8)
    H, W = I1.shape
    rs = np.ascontiguousarray(rand_stream, np.int32)
    minC = np.zeros((H, W), np.uint32)
    flow = np.zeros((2, H, W), np.float64)
    lib().fsgm_oracle_calc_cost_sgm_ng(_p(minC), _p(flow), _p(I1), _p(I2), W, H, int(P1), int(P2), _p(rs),
                                       C.c_int64(len(rs)))
    return minC, flow


def epi_fb_check(D1, pd0, nd, off, vMax, n, thr=2):
    """forward_backward_check of calc_cost_sgm.cpp:482-536 on bestD (before vz->disparity)."""
    D1 = np.ascontiguousarray(D1, np.uint32)
    H, W = D1.shape
    conf = np.zeros((H, W), np.uint8)
    D2 = np.zeros((H, W), np.uint32)
    lib().fsgm_oracle_epi_fb_check(_p(conf), _p(D2), _p(D1), W, H, _p(np.ascontiguousarray(pd0)),
                                   _p(np.ascontiguousarray(nd)), _p(np.ascontiguousarray(off)), C.c_double(vMax), int(n), int(thr))
    return conf, D2
