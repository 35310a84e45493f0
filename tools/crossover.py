import os, sys; sys.path.insert(0, '.')
import numpy as np, fsgm_amd
from fsgm_amd import synth, EpiPlan
from fsgm_amd._lib import STAGE_AGGREGATE, STAGE_WTA
# usage: crossover.py [batches] [WxHxD]   (default: the KITTI shape; BASELINE configs[1] is 320x240x64)
W, H, D = (int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (1242, 375, 128)
_, _, off = synth.epi_maps(W, H, "axis")
base = synth.cost_volume(W, H, D, seed=1, cmax=24)
Bs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8, 12, 16, 24, 40]
for paths in (8, 4):
    for B in Bs:
        r = []
        for strips in ("0",):
            plan = EpiPlan(W, H, D, B, paths=paths)
            plan.set_penalties(6, 64, 0.3)
            plan.upload_cost(0, base); plan.upload_offset(0, off)
            for f in range(1, B):
                plan.copy_cost(f, 0, 11 * f); plan.upload_offset(f, off)
            modes = (1, 3, 6, 2, 4, 5) if paths == 8 else (1, 2, 4, 5)
            for mode in modes:
                plan.set_agg_mode(mode)
                r.append((plan.kernel_name + ("+strips" if strips == "1" else ""), plan.time(STAGE_AGGREGATE | STAGE_WTA, 2, 6)))
            plan.close()
        print(f"{W}x{H}x{D} paths {paths} B {B}: " + "  ".join(f"{n} {ms:.3f}" for n, ms in r), flush=True)
