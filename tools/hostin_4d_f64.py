"""pca(4-D float64 numpy cube): channels uploaded by a thread one or two ahead of the fused float64 call (VIPMI_HOSTIN) against the
sequential loop: frame equality and time."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.psfsub import pca
rng = np.random.default_rng(0)
gc.collect(); gc.freeze()
for nch, n, N, k in ((39, 200, 256, 20), (6, 120, 256, 4)):
    cube = 7000 + 45 * rng.standard_normal((nch, n, N, N))
    ang = np.linspace(0, 100, n); res = {}
    for h in ("0", "1", "0", "1"):
        os.environ["VIPMI_HOSTIN"] = h
        pca(cube, ang, ncomp=k, verbose=False, check_memory=False); ts = []
        for _ in range(3):
            t0 = time.perf_counter(); out = pca(cube, ang, ncomp=k, verbose=False, check_memory=False); ts.append((time.perf_counter() - t0) * 1e3)
        res[h] = out
        print("%d x %d x %d^2 float64 k %d hostin %s: %.1f ms (min of 3)" % (nch, n, N, k, h, min(ts)), flush=True)
    print("   frames identical: %s" % np.array_equal(res["0"], res["1"], equal_nan=True))
