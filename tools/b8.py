"""8 frames, 8 paths, parallel sweeps (mode 3): for rocprofv3 --kernel-trace --stats.  usage: python tools/b8.py [B] [mode] [paths]"""
import sys; sys.path.insert(0, '.')
from fsgm_amd import synth, EpiPlan
from fsgm_amd._lib import STAGE_AGGREGATE, STAGE_WTA
W, H, D = 1242, 375, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
paths = int(sys.argv[3]) if len(sys.argv) > 3 else 8
_, _, off = synth.epi_maps(W, H, "axis")
base = synth.cost_volume(W, H, D, seed=1, cmax=24)
with EpiPlan(W, H, D, B, paths=paths) as plan:
    plan.set_penalties(6, 64, 0.3)
    plan.upload_cost(0, base); plan.upload_offset(0, off)
    for f in range(1, B):
        plan.copy_cost(f, 0, 11 * f); plan.upload_offset(f, off)
    plan.set_agg_mode(mode)
    t = min(plan.time(STAGE_AGGREGATE | STAGE_WTA, 3, 20) for _ in range(3))
    print(f"B {B} paths {paths} mode {mode}: {plan.kernel_name} {t:.3f} ms", flush=True)
