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

# ---- the MATLAB layers around the MEX files: pyramidal driver, post-processing chain,
# ---- epipolar maps.  Same caveat: this repo's oracle, kept as a regression anchor.
g0, g1 = synth.image_pair(40, 26, 10, seed=21)
R0 = np.stack([g0, 255 - g0, g0 // 3 + 80])
R1 = np.stack([g1, 255 - g1, g1 // 3 + 80])
mv, minC, lv = pyoracle.pyramidal_sgm(R0, R1, 3)
ext = {"pyr_mv": mv, "pyr_minC": minC, "pyr_lv2": lv[1], "pyr_lv3": lv[2]}
D1 = synth.vz_index_map(44, 30, 32, seed=22)
pd0p, ndp, offp = synth.epi_maps(44, 30, "general", seed=23)
f1, f2, disp = pyoracle.postprocess(D1, pd0p, ndp, offp / 8, 0.3, 33, 32)
ext.update(post_f1=f1, post_f2=f2, post_disp=disp)
F, Hm, epi, direction = synth.epi_geometry(36, 24, "contract")
P, n, o, r = pyoracle.epipolar_maps(F, Hm, epi, direction, 36, 24)
ext.update(geo_Pd0=P, geo_nd=n, geo_off=o, geo_rflow=r)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "oracle_layers.npz"), **ext)
print("wrote oracle_layers.npz")
