import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
cube, ang = synth_adi(400, 512, 0); ct = torch.from_numpy(cube).cuda()
ctx = B.get_context()
def prof(name, fn):
    fn(); torch.cuda.synchronize()
    ctx.set_option("timing", 1); ctx.reset_timers()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "%.2f ms" % (dt * 1e3), {s: round(ctx.stage_ms(s), 2) for s in ("scale", "gram", "eigh", "project", "derotate", "collapse") if ctx.stage_count(s)})
    ctx.set_option("timing", 0)
prof("trimmean", lambda: pca(ct, ang, ncomp=20, collapse="trimmean", verbose=False))
prof("cevr 0.9", lambda: pca(ct, ang, ncomp=0.9, verbose=False))
import cProfile, pstats
for nm, fn in (("trimmean", lambda: pca(ct, ang, ncomp=20, collapse="trimmean", verbose=False)), ("cevr", lambda: pca(ct, ang, ncomp=0.9, verbose=False))):
    pr = cProfile.Profile(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    print("==", nm); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
