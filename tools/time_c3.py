"""C3 (annular PCA, 400x512x512, 8 annuli, k=10): total and per-stage times of one call."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca_annular
cube, ang = synth_adi(400, 512, 0); ct = torch.from_numpy(cube).cuda()
f = lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, delta_rot=(0.1, 1), verbose=False).cpu()
f(); torch.cuda.synchronize()
ctx = B.get_context(); ctx.set_option("timing", 1); ctx.reset_timers()
t = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("C3 with stage timing: %.2f ms" % (dt * 1e3), {s: round(ctx.stage_ms(s), 2) for s in ("scale", "gram", "eigh", "project", "derotate", "collapse")})
ctx.set_option("timing", 0)
t = time.perf_counter()
for _ in range(3): f()
torch.cuda.synchronize(); print("C3: %.2f ms" % ((time.perf_counter() - t) / 3 * 1e3))
