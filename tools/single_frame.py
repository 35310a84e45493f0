"""Single-call regimes (1, 2 frames; 8 and 4 paths) through the line kernels: aggregation and WTA stage times.
usage: python tools/single_frame.py [WxHxD]"""
import sys; sys.path.insert(0, '.')
from fsgm_amd import synth, EpiPlan
from fsgm_amd._lib import STAGE_AGGREGATE, STAGE_WTA
W, H, D = (int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (1242, 375, 128)
_, _, off = synth.epi_maps(W, H, "axis")
base = synth.cost_volume(W, H, D, seed=1, cmax=24)
for paths in (8, 4):
    for B in (1, 2):
        with EpiPlan(W, H, D, B, paths=paths) as plan:
            plan.set_penalties(6, 64, 0.3)
            for f in range(B):
                plan.upload_cost(f, base); plan.upload_offset(f, off)
            plan.set_agg_mode(1)
            a = min(plan.time(STAGE_AGGREGATE, 3, 20) for _ in range(3))
            w = min(plan.time(STAGE_WTA, 3, 20) for _ in range(3))
            t = min(plan.time(STAGE_AGGREGATE | STAGE_WTA, 3, 20) for _ in range(3))
            print(f"{W}x{H}x{D} paths {paths} B {B}: {plan.kernel_name} aggregate {a:.3f} wta {w:.3f} both {t:.3f} ms", flush=True)
