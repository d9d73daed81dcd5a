"""BASELINE.json configs[4] feasibility: 2000x1024x1024, ncomp=50 on one GPU (exact decomposition)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
g = torch.Generator(device="cuda").manual_seed(0)
yy, xx = torch.meshgrid(torch.arange(N, device="cuda"), torch.arange(N, device="cuda"), indexing="ij")
env = torch.exp(-torch.sqrt((yy - N // 2) ** 2.0 + (xx - N // 2) ** 2.0) / (N / 8)).float()
modes = torch.randn((30, N, N), device="cuda", generator=g) * env
coef = torch.randn((n, 30), device="cuda", generator=g) * (2.0 ** (-torch.arange(30, device="cuda") / 3))
cube = torch.empty((n, N, N), device="cuda")
for i in range(0, n, 100):
    cube[i:i + 100] = torch.tensordot(coef[i:i + 100], modes, dims=1) + 3 * env + 0.2 * torch.randn((min(100, n - i), N, N), device="cuda", generator=g)
ang = np.linspace(0, 90, n)
ctx = B.get_context(); ctx.set_option("timing", 1)
for rep in range(2):
    ctx.reset_timers(); torch.cuda.synchronize(); t = time.perf_counter()
    fr = pca(cube, ang, ncomp=k, verbose=False, check_memory=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("rep", rep, "%.1f ms  %.0f frames/s" % (dt * 1e3, n / dt), {s: round(ctx.stage_ms(s), 2) for s in ("gram", "eigh", "project", "derotate", "collapse") if ctx.stage_count(s)},
          "sweeps", ctx.get_option("eigh_last_sweeps"), "finite", bool(torch.isfinite(fr).all()), "mem GB", round(torch.cuda.max_memory_allocated() / 1e9, 1))
