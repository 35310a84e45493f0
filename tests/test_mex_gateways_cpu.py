"""The mexFunction gateways (the reference's four MEX names and the pyramidal driver): they load, export mexFunction, and validate arguments loudly
(no GPU needed: validation happens before the library is asked to compute)."""
import ctypes
import os
import numpy as np
import pytest

from fsgm_amd import synth, _lib
from tests import mexharness as mh

GATEWAYS = ["calc_cost_sgm", "calc_pyd_cost_sgm", "calc_pyd_cost_sgm_ng", "calc_cost_sgm_ng", "fsgm_pyramidal_sgm",
            "fsgm_pyramidal_sgm_ng"]


@pytest.mark.parametrize("name", GATEWAYS)
def test_gateway_exports_mexfunction(name):
    mh.stub()
    lib = ctypes.CDLL(os.path.join(mh.MEXDIR, f"{name}.mexstub.so"))
    assert hasattr(lib, "mexFunction")


def _epi_args(W=32, H=24, D=16):
    I1, I2 = synth.image_pair(W, H, D)
    pd0, nd, off = synth.epi_maps(W, H)
    return [I1, I2, D, 0.3, pd0, nd, off, 6, 64]


def test_wrong_argument_count_and_classes():
    a = _epi_args()
    with pytest.raises(mh.MexError) as e:
        mh.call("calc_cost_sgm", 2, *a[:8])
    assert e.value.ident == "fsgm:nrhs"
    with pytest.raises(mh.MexError) as e:
        mh.call("calc_cost_sgm", 5, *a)
    assert e.value.ident == "fsgm:nlhs"
    b = list(a); b[0] = a[0].astype(np.float64)
    with pytest.raises(mh.MexError) as e:
        mh.call("calc_cost_sgm", 2, *b)
    assert e.value.ident == "fsgm:class"
    b = list(a); b[1] = a[1][:, :-1]
    with pytest.raises(mh.MexError) as e:
        mh.call("calc_cost_sgm", 2, *b)
    assert e.value.ident == "fsgm:size"
    b = list(a); b[4] = a[4][0]
    with pytest.raises(mh.MexError) as e:
        mh.call("calc_cost_sgm", 2, *b)
    assert e.value.ident == "fsgm:size"
    b = list(a); b[2] = 0
    with pytest.raises(mh.MexError) as e:
        mh.call("calc_cost_sgm", 2, *b)
    assert e.value.ident == "fsgm:range"


def test_pyd_gateway_validation():
    I1, I2 = synth.image_pair(20, 16, 16)
    mv = synth.hint_map(20, 16)
    with pytest.raises(mh.MexError) as e:                  # hint map smaller than the image
        mh.call("calc_pyd_cost_sgm", 3, I1, I2, mv[:, :10, :], 5, 5, 2, 0, 6, 32, 1, 2, 0)
    assert e.value.ident == "fsgm:size"
    with pytest.raises(mh.MexError) as e:                  # fractional window half size
        mh.call("calc_pyd_cost_sgm", 3, I1, I2, mv, 2.5, 5, 2, 0, 6, 32, 1, 2, 0)
    assert e.value.ident == "fsgm:range"
    with pytest.raises(mh.MexError) as e:
        mh.call("calc_pyd_cost_sgm_ng", 2, I1, I2, mv, 1, 2, 0, 6)
    assert e.value.ident == "fsgm:nrhs"


def test_pyramid_gateway_validation():
    I0, I1 = synth.image_pair(20, 16, 8)
    with pytest.raises(mh.MexError) as e:
        mh.call("fsgm_pyramidal_sgm", 2, I0)
    assert e.value.ident == "fsgm:nrhs"
    with pytest.raises(mh.MexError) as e:                  # two planes: neither gray nor RGB
        mh.call("fsgm_pyramidal_sgm", 2, np.stack([I0, I0]), np.stack([I1, I1]), 3)
    assert e.value.ident == "fsgm:size"
    with pytest.raises(mh.MexError) as e:
        mh.call("fsgm_pyramidal_sgm", 2, I0, I1[:, :-1], 3)
    assert e.value.ident == "fsgm:size"
    with pytest.raises(mh.MexError) as e:
        mh.call("fsgm_pyramidal_sgm", 2, I0, I1, 0)
    assert e.value.ident == "fsgm:range"
    with pytest.raises(mh.MexError) as e:                  # more outputs than mv, minC and one flow per level
        mh.call("fsgm_pyramidal_sgm", 6, I0, I1, 3)
    assert e.value.ident == "fsgm:nlhs"


def test_no_gpu_is_a_mex_error_not_a_fallback():
    if _lib.load().fsgm_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(mh.MexError) as e:
        mh.call("calc_cost_sgm", 2, *_epi_args())
    assert e.value.ident == "fsgm:hip" and "no HIP device" in str(e.value)
