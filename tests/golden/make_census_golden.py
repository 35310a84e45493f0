"""Generates tests/golden/census_ref_61x47.npz from the REFERENCE's own census()
(/root/reference/common.cpp:3-27, compiled unmodified into oracle/_ref/ by oracle/Makefile).
Run in the build container (the reference tree is not on the GPU box):
    python tests/golden/make_census_golden.py
The fixture holds data only: the input image and the census codes the reference produced."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fsgm_amd import synth          # noqa: E402
from oracle import pyoracle         # noqa: E402

img = synth.image_pair(61, 47, 16, seed=77)[0]
img[0, :5] = 0
img[-1, -5:] = 255
cen = pyoracle.ref_census(img)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "census_ref_61x47.npz"), img=img, cen=cen)
print("wrote census_ref_61x47.npz", cen.shape, hex(int(cen[20, 20])))
