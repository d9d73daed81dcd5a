"""pca(4-D float32 numpy cube) with the channel groups uploaded beside the PCA of the group before (VIPMI_HOSTIN) against
upload-then-call: frame equality and time at C4 size (39 x 200 x 256 x 256) and on a smaller cube."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.psfsub import pca
rng = np.random.default_rng(0)
gc.collect(); gc.freeze()
for nch, n, N, k, mpx in ((39, 200, 256, 20, None), (17, 150, 256, 5, 6), (8, 130, 384, 7, None)):
    cube = rng.standard_normal((nch, n, N, N), dtype=np.float32)
    cube += rng.standard_normal((nch, 1, N, N), dtype=np.float32) * 3
    ang = np.linspace(0, 100, n)
    res = {}
    for h in ("0", "1", "0", "1"):
        os.environ["VIPMI_HOSTIN"] = h
        pca(cube, ang, ncomp=k, mask_center_px=mpx, verbose=False, check_memory=False)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); out = pca(cube, ang, ncomp=k, mask_center_px=mpx, verbose=False, check_memory=False); ts.append((time.perf_counter() - t0) * 1e3)
        res[h] = out
        print("%d x %d x %d^2 k %d mask %s hostin %s: %.1f ms (min of 4)" % (nch, n, N, k, mpx, h, min(ts)), flush=True)
    print("   frames identical: %s (max |diff| %.1e)" % (np.array_equal(res["0"], res["1"], equal_nan=True), np.nanmax(np.abs(res["0"] - res["1"]))))
