"""fsgm_amd -- MI355X-native (gfx950, hand-written HIP) build of fSGM's matching-cost +
multi-path SGM aggregation hot path, behind the reference's MEX argument lists.

    from fsgm_amd import calc_cost_sgm            # mirrors calc_cost_sgm.cpp's mexFunction
    bestD, minC = calc_cost_sgm(I1, I2, dMax, vMax, pixelPosD0, normDir, offset, P1, P2)

Everything computes on the GPU through libfsgm_hip.so (C ABI in include/fsgm.h).
"""
from .epi import calc_cost_sgm, calc_cost_sgm_batch, EpiPlan, epipolar_maps, epipolar_sgm_of, epipolar_from_F, census, sgm, auto_pipeline  # noqa: F401
from .pyd import calc_pyd_cost_sgm, calc_pyd_cost_sgm_batch, PydPlan  # noqa: F401
from .pyramid import pyramidal_sgm, pyramidal_sgm_ng, pyramidal_sgm_batch, pyramidal_sgm_ng_batch, PyramidPlan, NgPyramidPlan  # noqa: F401
from .post import (speckle_filter, calc_disp_from_first, forward_backward_check, scanline_in_fill, vzInd2Disp, vmf,  # noqa: F401
                   epi_postprocess, PostPlan)
from .ng import calc_pyd_cost_sgm_ng, calc_cost_sgm_ng, calc_pyd_cost_sgm_ng_batch, calc_cost_sgm_ng_batch  # noqa: F401
from ._lib import FsgmError, load as load_library  # noqa: F401
