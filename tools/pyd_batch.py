import sys; sys.path.insert(0,'.')  # run from the repo root: PyramidPlan timing (full-window matcher), tools/prof_r04.sh
import numpy as np
from fsgm_amd import synth, PyramidPlan
W,H=1242,375
g0,g1=synth.image_pair(W,H,16,seed=2)
I0=np.stack([g0,255-g0,g0//2+40]); I1=np.stack([g1,255-g1,g1//2+40])
B=int(sys.argv[1]) if len(sys.argv)>1 else 8
with PyramidPlan(W,H,3,3,batch=B) as plan:
    for f in range(B): plan.upload(np.roll(I0,13*f,axis=2), np.roll(I1,13*f,axis=2), frame=f)
    print('batch',B,'ms per batch', plan.time(1,3))
