import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence
from oracle_lib import Oracle
from vp8oclenc_amd.driver import InterPathDriver
tot=0; bad=0; mx=0
for (W,H,seed,t,noise) in [(320,192,1,-1.0,4),(320,192,2,0.97,12),(640,368,3,0.93,8)]:
    s = SynthSequence(W,H,seed=seed,noise=noise)
    hip, ora = api.Vp8Hip(W,H,t), Oracle(W,H,t)
    dh, do = InterPathDriver(hip,W,H,ssim_target=t,check_ssim=False), InterPathDriver(ora,W,H,ssim_target=t,check_ssim=False)
    for f in range(6):
        y,u,v = s.frame(f)
        a,b = dh.encode_frame(y,u,v), do.encode_frame(y,u,v)
        if a is None: continue
        x,yv = a["MB_SSIM"].view(np.uint32).astype(np.int64), b["MB_SSIM"].view(np.uint32).astype(np.int64)
        tot += x.size; bad += int((x!=yv).sum()); mx = max(mx, int(np.abs(x-yv).max()))
    hip.close(); ora.close()
print("MB_SSIM values", tot, "bit-different", bad, "max ulp", mx)
