#!/usr/bin/env python3
"""Cost stage of calc_cost_sgm (census x2 + raw cost + 5x5 box mean) alone: ms per 1242x375x128 frame, fused kernel
and the two-kernel form (FSGM_COST_FUSED=0), on the survey's timing maps and on a random direction per pixel.
usage: tools/cost_stage.py [frames=64] [modes=fused,split] [kinds=axis,general] [iters=5]     (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fsgm_amd import synth, EpiPlan
from fsgm_amd._lib import STAGE_COST
W, H, D = 1242, 375, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
modes = (sys.argv[2] if len(sys.argv) > 2 else "fused,split").split(",")
kinds = (sys.argv[3] if len(sys.argv) > 3 else "axis,general").split(",")
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
I1, I2 = synth.image_pair(W, H, D, seed=3)
plan = EpiPlan(W, H, D, B, paths=8)
plan.set_penalties(6, 64, 0.3)
for kind in kinds:
    pd0, nd, off = synth.epi_maps(W, H, kind)
    for f in range(B):
        plan.upload(f, I1, I2, pd0, nd, off)
    for mode in modes:
        os.environ["FSGM_COST_FUSED"] = "1" if mode == "fused" else "0"
        ms = plan.time(STAGE_COST, 1, iters)
        print(f"{mode:5s} B {B} maps {kind:7s}: cost stage {ms / B:.4f} ms per frame", flush=True)
plan.close()
