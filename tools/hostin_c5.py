"""pca(numpy cube) at C5 size (2000 x 1024 x 1024, ncomp 50): host-input entry (Gram under the upload) against upload-then-call."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.psfsub import pca
rng = np.random.default_rng(0)
n, N = 2000, 1024
cube = rng.standard_normal((n, N, N), dtype=np.float32)
cube += (np.linspace(0, 3, n, dtype=np.float32)[:, None, None] * np.float32(0.5))
ang = np.linspace(0, 120, n)
res = {}
gc.collect(); gc.freeze()
for h in ("0", "1", "0", "1"):
    os.environ["VIPMI_HOSTIN"] = h
    t0 = time.perf_counter(); out = pca(cube, ang, ncomp=50, verbose=False, check_memory=False); dt = (time.perf_counter() - t0) * 1e3
    res[h] = out
    print("hostin %s: %.1f ms" % (h, dt), flush=True)
print("frames identical:", np.array_equal(res["0"], res["1"], equal_nan=True))
