"""Small batches, 8 paths: the sweep pipelines side by side (modes 3, 6, 2).  usage: python tools/b8.py [B,B,...] [modes] [WxHxD]"""
import sys; sys.path.insert(0, '.')
from fsgm_amd import synth, EpiPlan
from fsgm_amd._lib import STAGE_AGGREGATE, STAGE_WTA
W, H, D = (int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (1242, 375, 128)
Bs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8]
modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 6, 2]
_, _, off = synth.epi_maps(W, H, "axis")
base = synth.cost_volume(W, H, D, seed=1, cmax=24)
for B in Bs:
    with EpiPlan(W, H, D, B, paths=8) as plan:
        plan.set_penalties(6, 64, 0.3)
        plan.upload_cost(0, base); plan.upload_offset(0, off)
        for f in range(1, B):
            plan.copy_cost(f, 0, 11 * f); plan.upload_offset(f, off)
        r = []
        for mode in modes:
            plan.set_agg_mode(mode)
            r.append((plan.kernel_name, min(plan.time(STAGE_AGGREGATE | STAGE_WTA, 3, 20) for _ in range(3))))
        print(f"{W}x{H}x{D} B {B}: " + "  ".join(f"{n} {t:.3f}" for n, t in r), flush=True)
