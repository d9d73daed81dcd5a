"""C4 (39 channels x 200 x 256 x 256, ncomp 20): total and per-stage times of one frame-only call."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
cubes = np.stack([synth_adi(200, 256, s)[0] for s in range(39)]); ang = np.linspace(0, 90, 200)
ct = torch.from_numpy(cubes).cuda()
f = lambda: pca(ct, ang, ncomp=20, verbose=False, check_memory=False).cpu()
f(); torch.cuda.synchronize()
ctx = B.get_context(); ctx.set_option("timing", 1); ctx.reset_timers()
t = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("C4 with stage timing: %.2f ms" % (dt * 1e3), {s: round(ctx.stage_ms(s), 2) for s in ("gram", "eigh", "project", "derotate", "collapse", "k_rot_s1", "k_rot_s2", "k_rot_s3")})
ctx.set_option("timing", 0)
t = time.perf_counter()
for _ in range(3): f()
torch.cuda.synchronize(); print("C4: %.2f ms" % ((time.perf_counter() - t) / 3 * 1e3))
