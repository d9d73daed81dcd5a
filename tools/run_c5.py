"""BASELINE.json configs[4] feasibility: 2000x1024x1024, ncomp=50 on one GPU (exact decomposition)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
from vip_amd.synth import synth_adi_device
cube, ang = synth_adi_device(n, N, seed=0)
ctx = B.get_context(); ctx.set_option("timing", 1)
for rep in range(2):
    ctx.reset_timers(); torch.cuda.synchronize(); t = time.perf_counter()
    fr = pca(cube, ang, ncomp=k, verbose=False, check_memory=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("rep", rep, "%.1f ms  %.0f frames/s" % (dt * 1e3, n / dt), {s: round(ctx.stage_ms(s), 2) for s in ("gram", "eigh", "project", "derotate", "collapse") if ctx.stage_count(s)},
          "sweeps", ctx.get_option("eigh_last_sweeps"), "finite", bool(torch.isfinite(fr).all()), "mem GB", round(torch.cuda.max_memory_allocated() / 1e9, 1))
print("fast path of the eigensolver (products, rounds, locked, reason):",
      [ctx.get_option("eigh_fast_last_" + s) for s in ("products", "rounds", "locked", "reason")])
