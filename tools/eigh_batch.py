import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
ctx = B.get_context(); ctx.set_option("timing", 1)
for n, batch in ((400,1),(400,4),(400,16),(200,1),(200,36),(100,1),(100,64)):
    M = rng.standard_normal((n, 2*n)); M[:, :5] *= 30
    G = M @ M.T
    Gs = torch.from_numpy(np.stack([G]*batch)).cuda()
    for r in range(2):
        ctx.reset_timers()
        g2 = Gs.clone()
        torch.cuda.synchronize(); t=time.perf_counter()
        B.eigh(g2)
        torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(n, batch, "eigh ms", round(ctx.stage_ms("eigh"),3), "wall", round(dt*1e3,3), "sweeps", ctx.get_option("eigh_last_sweeps"))
