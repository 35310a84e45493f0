"""Drives the mexFunction gateways (fsgm_amd/mex/*.mexstub.so) the way MATLAB would, through the
test stub of the MEX API (tests/mexstub).  numpy (H, W) C-order <-> MATLAB W x H column-major."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEXDIR = os.path.join(ROOT, "fsgm_amd", "mex")
_CLS = {np.dtype(np.float64): 6, np.dtype(np.uint8): 9, np.dtype(np.uint32): 13}
_DT = {6: np.float64, 9: np.uint8, 13: np.uint32}
_stub = None


class MexError(RuntimeError):
    def __init__(self, ident, msg):
        super().__init__(f"{ident}: {msg}")
        self.ident = ident


def stub():
    global _stub
    if _stub is None:
        s = C.CDLL(os.path.join(MEXDIR, "libmexstub.so"), mode=C.RTLD_GLOBAL)
        s.mxCreateNumericArray.restype = C.c_void_p
        s.mxCreateNumericArray.argtypes = [C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_int]
        s.mxCreateDoubleScalar.restype = C.c_void_p
        s.mxCreateDoubleScalar.argtypes = [C.c_double]
        s.mxDestroyArray.argtypes = [C.c_void_p]
        s.mxGetData.restype = C.c_void_p
        s.mxGetData.argtypes = [C.c_void_p]
        s.mxGetNumberOfDimensions.restype = C.c_size_t
        s.mxGetNumberOfDimensions.argtypes = [C.c_void_p]
        s.mxGetDimensions.restype = C.POINTER(C.c_size_t)
        s.mxGetDimensions.argtypes = [C.c_void_p]
        s.mxGetClassID.argtypes = [C.c_void_p]
        s.mexstub_call.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        s.mexstub_last_error_id.restype = C.c_char_p
        s.mexstub_last_error_msg.restype = C.c_char_p
        s.mexstub_printed.restype = C.c_char_p
        _stub = s
    return _stub


def to_mx(a):
    s = stub()
    if np.isscalar(a):
        return s.mxCreateDoubleScalar(float(a))
    a = np.ascontiguousarray(a)
    dims = tuple(reversed(a.shape))                       # C-order (.., H, W) -> MATLAB W x H x ..
    arr = (C.c_size_t * len(dims))(*dims)
    m = s.mxCreateNumericArray(len(dims), arr, _CLS[a.dtype], 0)
    C.memmove(s.mxGetData(m), a.ctypes.data, a.nbytes)
    return m


def from_mx(m):
    s = stub()
    nd = s.mxGetNumberOfDimensions(m)
    dims = [s.mxGetDimensions(m)[i] for i in range(nd)]
    dt = _DT[s.mxGetClassID(m)]
    n = int(np.prod(dims))
    out = np.empty(tuple(reversed(dims)), dt)
    C.memmove(out.ctypes.data, s.mxGetData(m), n * np.dtype(dt).itemsize)
    return out


def call(name, nlhs, *args):
    """outputs = call('calc_cost_sgm', 2, I1, I2, 64, 0.3, ...) ; raises MexError on mexErrMsgIdAndTxt"""
    s = stub()
    lib = C.CDLL(os.path.join(MEXDIR, f"{name}.mexstub.so"))
    fn = C.cast(lib.mexFunction, C.c_void_p)
    prhs = (C.c_void_p * max(1, len(args)))(*[to_mx(a) for a in args])
    plhs = (C.c_void_p * max(1, nlhs, 20))()
    rc = s.mexstub_call(fn, nlhs, plhs, len(args), prhs)
    printed = s.mexstub_printed().decode()
    try:
        if rc:
            raise MexError(s.mexstub_last_error_id().decode(), s.mexstub_last_error_msg().decode())
        outs = [from_mx(plhs[i]) for i in range(max(1, nlhs))]
    finally:
        for i in range(len(args)):
            s.mxDestroyArray(prhs[i])
    return outs, printed
