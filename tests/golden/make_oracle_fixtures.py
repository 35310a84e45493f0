"""Generates the oracle self-regression fixture oracle_epi_48x36x16.npz.  These vectors come
from THIS REPO'S oracle, not from the reference (whose MEX sources cannot be built here without
MATLAB's mex.h); they only guard the oracle against accidental edits."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fsgm_amd import synth          # noqa: E402
from oracle import pyoracle         # noqa: E402

W, H, D = 48, 36, 16
I1, I2 = synth.image_pair(W, H, D, seed=11)
pd0, nd, off = synth.epi_maps(W, H, "general", seed=12)
out = {}
for paths in (4, 8):
    bd, mc = pyoracle.calc_cost_sgm(I1, I2, D, 0.3, pd0, nd, off, 6, 64, paths)
    out[f"bestD{paths}"], out[f"minC{paths}"] = bd, mc
np.savez_compressed(os.path.join(os.path.dirname(__file__), "oracle_epi_48x36x16.npz"), **out)
print("wrote oracle_epi_48x36x16.npz")
