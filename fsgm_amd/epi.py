"""Host-side mirror of the reference's calc_cost_sgm MEX interface (calc_cost_sgm.cpp:539-598,
called from epipolar_sgm_of.m:45), on top of the C ABI.

Array convention = the MEX's memory order (include/fsgm.h): what MATLAB passes as a
width x height column-major matrix is here a C-contiguous numpy array of shape (height, width);
width x height x 2 maps are (2, height, width).
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import EpiParams, EpiIn, EpiOut, check, ptr


def _u8img(a, name):
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise TypeError(f"{name} must be a 2-D uint8 array (got {a.dtype}, ndim {a.ndim})")
    return np.ascontiguousarray(a)


def _f64(a, shape, name):
    a = np.asarray(a)
    if a.dtype != np.float64 or a.shape != shape:
        raise TypeError(f"{name} must be float64 of shape {shape} (got {a.dtype} {a.shape})")
    return np.ascontiguousarray(a)


def _params(paths, subpixel, vz_to_disp, device, fb_check=0):
    p = EpiParams()
    p.paths, p.subpixel, p.vz_to_disp, p.device = int(paths), int(subpixel), int(vz_to_disp), int(device)
    p.fb_check = int(fb_check)
    return p


def auto_pipeline(width, height, dMax, batch, paths=4, P1=6, P2=64, cmax=24, cus=256):
    """Which aggregation pipeline auto mode takes for `batch` frames of this shape (fsgm_epi_auto_pipeline): a pure function of
    its arguments and the FSGM_EPI_* environment, no device needed."""
    return _lib.load().fsgm_epi_auto_pipeline(int(width), int(height), int(dMax), int(batch), int(paths), int(P1), int(P2), int(cmax), int(cus)).decode()


def calc_cost_sgm_batch(frames, dMax, vMax, P1, P2, *, paths=4, subpixel=1, vz_to_disp=1, device=0,
                        return_volumes=False, fb_check=0, devices=None):
    """frames: list of (I1, I2, pixelPosD0, normDir, offset) of one shape, processed concurrently.

    devices: a device list (sequence of HIP ordinals, or text "0,1,2" like FSGM_DEVICES) -- frame i runs on
    devices[i % len(devices)], one host thread per entry inside the library, no collective
    (fsgm_calc_cost_sgm_batch_devices_host); `device` is ignored then.  Results do not depend on the list."""
    lib = _lib.load()
    n = len(frames)
    if n == 0:
        return []
    ins, outs, keep, res = (EpiIn * n)(), (EpiOut * n)(), [], []
    H, W = np.asarray(frames[0][0]).shape
    D = int(dMax)
    for i, (I1, I2, pd0, nd, off) in enumerate(frames):
        I1, I2 = _u8img(I1, "I1"), _u8img(I2, "I2")
        if I1.shape != (H, W) or I2.shape != (H, W):
            raise ValueError("all images of a batch must share one shape")
        pd0 = _f64(pd0, (2, H, W), "pixelPosD0")
        nd = _f64(nd, (2, H, W), "normlizeDirection")
        off = _f64(off, (H, W), "offsetFromPosD0")
        bestD = np.zeros((H, W), np.uint32)
        minC = np.zeros((H, W), np.uint32)
        Cv = np.zeros((H, W, D), np.uint8) if return_volumes else None
        Sv = np.zeros((H, W, D), np.uint32) if return_volumes else None
        keep.append((I1, I2, pd0, nd, off))
        e = ins[i]
        e.I1, e.I2, e.width, e.height, e.dMax, e.vMax = ptr(I1), ptr(I2), W, H, D, float(vMax)
        e.pixelPosD0, e.normDir, e.offset, e.P1, e.P2 = ptr(pd0), ptr(nd), ptr(off), int(P1), int(P2)
        o = outs[i]
        conf = np.zeros((H, W), np.uint8) if fb_check else None
        bestD2 = np.zeros((H, W), np.uint32) if fb_check else None
        o.bestD, o.minC, o.C, o.S, o.conf, o.bestD2 = ptr(bestD), ptr(minC), ptr(Cv), ptr(Sv), ptr(conf), ptr(bestD2)
        r = (bestD, minC, Cv, Sv) if return_volumes else (bestD, minC)
        res.append(r + (conf, bestD2) if fb_check else r)
    prm = _params(paths, subpixel, vz_to_disp, device, fb_check)
    if devices is not None:
        nd, darr = _lib.device_array(devices)
        check(lib.fsgm_calc_cost_sgm_batch_devices_host(n, ins, outs, C.byref(prm), nd, darr))
    else:
        check(lib.fsgm_calc_cost_sgm_batch_host(n, ins, outs, C.byref(prm)))
    return res


def calc_cost_sgm(I1, I2, dMax, vMax, pixelPosD0, normlizeDirection, offsetFromPosD0, P1, P2, *,
                  paths=4, subpixel=1, vz_to_disp=1, device=0, return_volumes=False, fb_check=0):
    """[bestD, minC] = calc_cost_sgm(I1, I2, dMax, vMax, pixelPosD0, normlizeDirection,
    offsetFromPosD0, P1, P2)  -- same argument order and meaning as the MEX.

    Keyword arguments are the reference's compile-time switches (defaults = as shipped).
    fb_check=1 additionally returns (conf, bestD2): the forward-backward check the reference has
    commented out (calc_cost_sgm.cpp:482-536, :589-590).
    """
    return calc_cost_sgm_batch([(I1, I2, pixelPosD0, normlizeDirection, offsetFromPosD0)], dMax, vMax, P1, P2,
                               paths=paths, subpixel=subpixel, vz_to_disp=vz_to_disp, device=device,
                               return_volumes=return_volumes, fb_check=fb_check)[0]


def sgm(Cvol, P1=7, P2=100, *, paths=4, device=0, return_sum=False):
    """[bestD, minC] = sgm(C, P1, P2): sgm.m's call shape (sgm.m:1, defaults :4-10; test.m:36 passes 6, 64) on the MEX's
    aggregation and WTA.  C: (height, width, dMax) uint8.  bestD = disparity index * 256 (MEX parabola).  MEX semantics
    (mod-256 path arithmetic, path start stores minimum 0), not sgm.m's saturating ones: see include/fsgm.h."""
    lib = _lib.load()
    Cvol = np.ascontiguousarray(Cvol)
    if Cvol.dtype != np.uint8 or Cvol.ndim != 3:
        raise TypeError("C must be a (height, width, dMax) uint8 array")
    H, W, D = Cvol.shape
    bestD, minC = np.empty((H, W), np.uint32), np.empty((H, W), np.uint32)
    S = np.empty((H, W, D), np.uint32) if return_sum else None
    check(lib.fsgm_sgm_host(ptr(Cvol), W, H, D, int(P1), int(P2), int(paths), ptr(bestD), ptr(minC), ptr(S), int(device)))
    return (bestD, minC, S) if return_sum else (bestD, minC)


def census(img, *, device=0):
    """census(img) of common.cpp:3-27 on the device: (height, width) uint8 -> uint32 codes."""
    lib = _lib.load()
    img = _u8img(img, "img")
    H, W = img.shape
    cen = np.empty((H, W), np.uint32)
    check(lib.fsgm_census_host(ptr(img), W, H, ptr(cen), int(device)))
    return cen


class EpiPlan:
    """Device-resident plan: `batch` frames of width x height x dMax stay in HBM across calls."""

    def __init__(self, width, height, dMax, batch=1, *, paths=4, subpixel=1, vz_to_disp=1, device=0, fb_check=0):
        self.lib = _lib.load()
        self.W, self.H, self.D, self.batch, self.paths = int(width), int(height), int(dMax), int(batch), int(paths)
        self._h = C.c_void_p()
        prm = _params(paths, subpixel, vz_to_disp, device, fb_check)
        check(self.lib.fsgm_epi_plan_create(C.byref(self._h), self.W, self.H, self.D, self.batch, C.byref(prm)))

    def close(self):
        if self._h:
            self.lib.fsgm_epi_plan_destroy(self._h)
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

    def set_penalties(self, P1, P2, vMax=0.3):
        check(self.lib.fsgm_epi_plan_set_penalties(self._h, int(P1), int(P2), float(vMax)))

    def set_agg_mode(self, mode):
        """0 auto, 1 per-direction line kernels, 2 fused pipeline when eligible, 3 parallel sweeps (8 paths) when eligible,
        4 band sweeps (all four paths of a pass in one sweep, one workgroup per frame: very large batches), 5 the same with the
        bands of a frame as workgroups of their own that hand over while they run (chained: batches from ~100 frames), 6 the
        parallel sweeps meeting in the middle (each finishes the other's half of the rows with the WTA inside: 10-25 frames)."""
        check(self.lib.fsgm_epi_plan_set_agg_mode(self._h, int(mode)))

    def upload(self, frame, I1, I2, pd0, nd, off):
        I1, I2 = _u8img(I1, "I1"), _u8img(I2, "I2")
        pd0 = _f64(pd0, (2, self.H, self.W), "pixelPosD0")
        nd = _f64(nd, (2, self.H, self.W), "normlizeDirection")
        off = _f64(off, (self.H, self.W), "offsetFromPosD0")
        check(self.lib.fsgm_epi_plan_upload(self._h, frame, ptr(I1), ptr(I2), ptr(pd0), ptr(nd), ptr(off)))

    def upload_cost(self, frame, Cvol):
        Cvol = np.ascontiguousarray(Cvol)
        if Cvol.dtype != np.uint8 or Cvol.shape != (self.H, self.W, self.D):
            raise TypeError(f"C must be uint8 of shape {(self.H, self.W, self.D)}")
        check(self.lib.fsgm_epi_plan_upload_cost(self._h, frame, ptr(Cvol)))

    def copy_cost(self, dst, src, roll_cols=0):
        """Resident cost volume of frame dst <- frame src, columns rotated by roll_cols (device to device)."""
        check(self.lib.fsgm_epi_plan_copy_cost(self._h, int(dst), int(src), int(roll_cols)))

    def upload_offset(self, frame, off):
        off = _f64(off, (self.H, self.W), "offsetFromPosD0")
        check(self.lib.fsgm_epi_plan_upload_offset(self._h, frame, ptr(off)))

    def run(self, stages=_lib.STAGE_ALL):
        check(self.lib.fsgm_epi_plan_run(self._h, int(stages)))

    def sync(self):
        check(self.lib.fsgm_epi_plan_sync(self._h))

    def download(self, frame):
        bestD = np.empty((self.H, self.W), np.uint32)
        minC = np.empty((self.H, self.W), np.uint32)
        check(self.lib.fsgm_epi_plan_download(self._h, frame, ptr(bestD), ptr(minC)))
        return bestD, minC

    def download_fb(self, frame):
        """(conf, bestD2) of the forward-backward check (plans created with fb_check=1)."""
        conf = np.empty((self.H, self.W), np.uint8)
        bestD2 = np.empty((self.H, self.W), np.uint32)
        check(self.lib.fsgm_epi_plan_download_fb(self._h, frame, ptr(conf), ptr(bestD2)))
        return conf, bestD2

    def download_cost(self, frame):
        Cv = np.empty((self.H, self.W, self.D), np.uint8)
        check(self.lib.fsgm_epi_plan_download_cost(self._h, frame, ptr(Cv)))
        return Cv

    def download_census(self, frame):
        """(cen1, cen2): the census codes FSGM_STAGE_COST computed for the two images (debug tap)."""
        c1 = np.empty((self.H, self.W), np.uint32)
        c2 = np.empty((self.H, self.W), np.uint32)
        check(self.lib.fsgm_epi_plan_download_census(self._h, frame, ptr(c1), ptr(c2)))
        return c1, c2

    def download_sum(self, frame):
        S = np.empty((self.H, self.W, self.D), np.uint32)
        check(self.lib.fsgm_epi_plan_download_sum(self._h, frame, ptr(S)))
        return S

    def time(self, stages, warmup=2, iters=10):
        ms = C.c_float()
        check(self.lib.fsgm_epi_plan_time(self._h, int(stages), int(warmup), int(iters), C.byref(ms)))
        return float(ms.value)

    @property
    def kernel_name(self):
        return self.lib.fsgm_epi_plan_kernel_name(self._h).decode()


# ---------------------------------------------------------------------------------------------
# epipolar driver, dense half (epipolar_geometry.m:99-115, rotation_motion.m, epipolar_sgm_of.m:33-51)
# ---------------------------------------------------------------------------------------------
class EpiGeometry(C.Structure):
    """What the sparse half of epipolar_geometry.m (:30-96, not built here) hands over."""
    _fields_ = [("F", C.c_double * 9), ("H", C.c_double * 9), ("epipole", C.c_double * 2), ("direction", C.c_int32)]


def _geometry(F, H, epipole, direction):
    F, H = np.asarray(F, np.float64), np.asarray(H, np.float64)
    if F.shape != (3, 3) or H.shape != (3, 3):
        raise TypeError("F and H must be 3x3 matrices")
    g = EpiGeometry()
    g.F[:] = list(F.reshape(-1))
    g.H[:] = list(H.reshape(-1))
    g.epipole[:] = [float(epipole[0]), float(epipole[1])]
    g.direction = int(bool(direction))
    return g


def _bind_driver(lib):
    if getattr(lib, "_epi_driver_bound", False):
        return
    vp, i32 = C.c_void_p, C.c_int32
    lib.fsgm_epipolar_maps_host.argtypes = [C.POINTER(EpiGeometry), i32, i32, vp, vp, vp, vp, i32]
    lib.fsgm_epipolar_sgm_of_host.argtypes = [vp, vp, i32, i32, i32, C.POINTER(EpiGeometry), i32, C.c_double,
                                              C.POINTER(_lib.EpiParams), vp, vp]
    lib._epi_driver_bound = True


def epipolar_from_F(F, K, pts1=None, pts2=None, inliers=None):
    """epipolar_geometry.m:40-96 on the host: (H, epipole, direction, ambiguous) from the fundamental matrix F, the intrinsics K
    and the matched points (n x 2 arrays of (x, y), MATLAB's 1-based pixel coordinates; inliers: boolean mask or None = all).
    What epipolar_maps / epipolar_sgm_of take next.  The SURF + LMedS estimate of F itself (:130-149) is toolbox code, not built."""
    lib = _lib.load()
    _bind_driver(lib)
    lib.fsgm_epipolar_from_F.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(EpiGeometry), C.POINTER(C.c_int32)]
    F, K = np.ascontiguousarray(F, np.float64), np.ascontiguousarray(K, np.float64)
    if F.shape != (3, 3) or K.shape != (3, 3):
        raise TypeError("F and K must be 3x3 matrices")
    n = 0 if pts1 is None else len(pts1)
    p1 = np.ascontiguousarray(pts1, np.float64).reshape(n, 2) if n else None
    p2 = np.ascontiguousarray(pts2, np.float64).reshape(n, 2) if n else None
    inl = np.ascontiguousarray(inliers, np.uint8) if inliers is not None else None
    if inl is not None and inl.shape != (n,):
        raise ValueError("inliers must have one entry per match")
    g, amb = EpiGeometry(), C.c_int32()
    check(lib.fsgm_epipolar_from_F(ptr(F), ptr(K), n, ptr(p1), ptr(p2), ptr(inl), C.byref(g), C.byref(amb)))
    return np.array(g.H[:]).reshape(3, 3), (g.epipole[0], g.epipole[1]), int(g.direction), bool(amb.value)


def epipolar_maps(F, H, epipole, direction, width, height, *, device=0):
    """(PrefD0, NormlizeDirection, Offset, Rflow) of epipolar_geometry.m:99-115 from F, H = K*R/K, the
    epipole in image 2 and the expansion/contraction flag."""
    lib = _lib.load()
    _bind_driver(lib)
    g = _geometry(F, H, epipole, direction)
    Wd, Hd = int(width), int(height)
    Pd0, nd, rflow = (np.empty((2, Hd, Wd), np.float64) for _ in range(3))
    off = np.empty((Hd, Wd), np.float64)
    check(lib.fsgm_epipolar_maps_host(C.byref(g), Wd, Hd, ptr(Pd0), ptr(nd), ptr(off), ptr(rflow), int(device)))
    return Pd0, nd, off, rflow


def epipolar_sgm_of(I0, I1, F, H, epipole, direction, dMax=64, vMax=0.3, *, paths=4, device=0):
    """[flow, minC] = epipolar_sgm_of(I0, I1, K, dMax, vMax) from epipolar_sgm_of.m:33 on, with the sparse
    geometry (F, H, epipole, direction) given.  I0/I1: (height, width) or (3, height, width) uint8;
    flow: (3, height, width) float64, third plane 1."""
    lib = _lib.load()
    _bind_driver(lib)
    I0, I1 = np.ascontiguousarray(I0), np.ascontiguousarray(I1)
    if I0.dtype != np.uint8 or I1.dtype != np.uint8 or I0.shape != I1.shape:
        raise TypeError("I0/I1 must be uint8 images of one shape")
    if not (I0.ndim == 2 or (I0.ndim == 3 and I0.shape[0] == 3)):
        raise TypeError("images must be (height, width) or (3, height, width)")
    Hd, Wd = I0.shape[-2:]
    g = _geometry(F, H, epipole, direction)
    prm = _params(paths, 1, 1, device, 0)
    flow = np.empty((3, Hd, Wd), np.float64)
    minC = np.empty((Hd, Wd), np.uint32)
    check(lib.fsgm_epipolar_sgm_of_host(ptr(I0), ptr(I1), Wd, Hd, 1 if I0.ndim == 2 else 3, C.byref(g), int(dMax), float(vMax),
                                        C.byref(prm), ptr(flow), ptr(minC)))
    return flow, minC
