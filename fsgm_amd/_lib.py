"""ctypes binding of libfsgm_hip.so (the C ABI declared in include/fsgm.h).

The library is the product; this module only loads it.  There is no CPU fallback: if the
shared object is missing, importing the compute wrappers raises with the build command.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FSGM_LIB_PATH: developer override to A/B two builds of the library in one session
LIB_PATH = os.environ.get("FSGM_LIB_PATH") or os.path.join(_HERE, "libfsgm_hip.so")

FSGM_OK = 0
STAGE_COST, STAGE_AGGREGATE, STAGE_WTA, STAGE_ALL = 1, 2, 4, 7


class FsgmError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"fsgm status {status}: {msg}")
        self.status = status


class EpiParams(C.Structure):
    _fields_ = [("paths", C.c_int32), ("subpixel", C.c_int32), ("vz_to_disp", C.c_int32), ("device", C.c_int32),
                ("fb_check", C.c_int32)]


class EpiIn(C.Structure):
    _fields_ = [("I1", C.c_void_p), ("I2", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("dMax", C.c_int32), ("vMax", C.c_double), ("pixelPosD0", C.c_void_p),
                ("normDir", C.c_void_p), ("offset", C.c_void_p), ("P1", C.c_int32), ("P2", C.c_int32)]


class EpiOut(C.Structure):
    _fields_ = [("bestD", C.c_void_p), ("minC", C.c_void_p), ("C", C.c_void_p), ("S", C.c_void_p),
                ("conf", C.c_void_p), ("bestD2", C.c_void_p)]


_lib = None


def load():
    """Load libfsgm_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first (`make` at the repo root, or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). fsgm_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int32, C.POINTER(C.c_float)
    lib.fsgm_last_error.restype = C.c_char_p
    lib.fsgm_device_count.restype = C.c_int
    lib.fsgm_device_arch.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    lib.fsgm_shutdown.restype = None
    lib.fsgm_epi_params_default.restype = EpiParams
    lib.fsgm_calc_cost_sgm_host.argtypes = [C.POINTER(EpiIn), C.POINTER(EpiOut), C.POINTER(EpiParams)]
    lib.fsgm_calc_cost_sgm_batch_host.argtypes = [i32, C.POINTER(EpiIn), C.POINTER(EpiOut), C.POINTER(EpiParams)]
    i32p = C.POINTER(C.c_int32)
    lib.fsgm_calc_cost_sgm_batch_devices_host.argtypes = [i32, C.POINTER(EpiIn), C.POINTER(EpiOut), C.POINTER(EpiParams), i32, i32p]
    lib.fsgm_parse_device_list.argtypes = [C.c_char_p, i32p, i32]
    lib.fsgm_parse_device_list.restype = i32
    lib.fsgm_shard_frames.argtypes = [i32, i32, i32, i32p, i32p]
    lib.fsgm_shard_frames.restype = None
    lib.fsgm_epi_plan_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, C.POINTER(EpiParams)]
    lib.fsgm_epi_plan_destroy.argtypes = [vp]
    lib.fsgm_epi_plan_destroy.restype = None
    lib.fsgm_epi_plan_set_penalties.argtypes = [vp, i32, i32, C.c_double]
    lib.fsgm_epi_plan_set_agg_mode.argtypes = [vp, i32]
    lib.fsgm_epi_plan_upload.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.fsgm_epi_plan_upload_cost.argtypes = [vp, i32, vp]
    lib.fsgm_epi_plan_upload_offset.argtypes = [vp, i32, vp]
    lib.fsgm_epi_plan_copy_cost.argtypes = [vp, i32, i32, i32]
    lib.fsgm_epi_plan_run.argtypes = [vp, i32]
    lib.fsgm_epi_plan_sync.argtypes = [vp]
    lib.fsgm_epi_plan_download.argtypes = [vp, i32, vp, vp]
    lib.fsgm_epi_plan_download_cost.argtypes = [vp, i32, vp]
    lib.fsgm_epi_plan_download_fb.argtypes = [vp, i32, vp, vp]
    lib.fsgm_epi_plan_download_sum.argtypes = [vp, i32, vp]
    lib.fsgm_epi_plan_download_census.argtypes = [vp, i32, vp, vp]
    lib.fsgm_census_host.argtypes = [vp, i32, i32, vp, i32]
    lib.fsgm_sgm_host.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32]
    lib.fsgm_epi_plan_time.argtypes = [vp, i32, i32, i32, f32p]
    lib.fsgm_epi_plan_stream.argtypes = [vp]
    lib.fsgm_epi_plan_stream.restype = vp
    lib.fsgm_epi_plan_kernel_name.argtypes = [vp]
    lib.fsgm_epi_plan_kernel_name.restype = C.c_char_p
    lib.fsgm_epi_auto_pipeline.argtypes = [i32] * 9
    lib.fsgm_epi_auto_pipeline.restype = C.c_char_p
    lib.fsgm_measure_copy_bandwidth.argtypes = [i32, C.c_size_t, i32, C.POINTER(C.c_double)]
    lib.fsgm_measure_copy_bandwidth2.argtypes = [i32, C.c_size_t, i32, i32, C.POINTER(C.c_double)]
    _lib = lib
    return lib


def check(status):
    if status != FSGM_OK:
        raise FsgmError(status, load().fsgm_last_error().decode())


def device_array(devices):
    """(count, int32 array) of a device list given as a sequence of ordinals or as text "0,1,2" (FSGM_DEVICES' format)."""
    if isinstance(devices, str):
        buf = (C.c_int32 * 1024)()
        n = load().fsgm_parse_device_list(devices.encode(), buf, 1024)
        if n <= 0:
            raise ValueError(f"not a device list: {devices!r}")
        return n, buf
    devs = [int(d) for d in devices]
    if not devs:
        raise ValueError("empty device list")
    return len(devs), (C.c_int32 * len(devs))(*devs)


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)
