"""Times ADI+mSDI double / single pass on a C4-sized IFS cube (39 channels x 200 frames x 256 x 256)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
z = int(sys.argv[1]) if len(sys.argv) > 1 else 39
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cube = np.stack([synth_adi(n, N, s)[0] for s in range(z)])
ang = np.linspace(0, 90, n)
sc = np.linspace(1.0, 1.3, z)[::-1].copy()
ct = torch.from_numpy(cube).cuda()
for mode, nc in (("double", (3, 10)),):
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        fr = pca(ct, ang, scale_list=sc, adimsdi=mode, ncomp=nc, verbose=False, check_memory=False)
        torch.cuda.synchronize()
        print(mode, nc, "%.1f ms" % ((time.perf_counter() - t) * 1e3), bool(torch.isfinite(fr).all()), "peak GB %.1f" % (torch.cuda.max_memory_allocated() / 1e9))
from vip_amd import backend as B
ctx = B.get_context(); ctx.set_option("timing", 1); ctx.reset_timers()
torch.cuda.synchronize(); t = time.perf_counter()
fr = pca(ct, ang, scale_list=sc, adimsdi="double", ncomp=(3, 10), verbose=False, check_memory=False)
torch.cuda.synchronize()
print("with stage timing: %.1f ms" % ((time.perf_counter() - t) * 1e3), {s: (round(ctx.stage_ms(s), 2), ctx.stage_count(s)) for s in ("scale", "gram", "eigh", "project", "derotate", "collapse")})
