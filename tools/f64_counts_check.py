import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_cpu as O
from vip_amd.psfsub import pca
g = np.load("tests/golden/g28_f64_counts.npz")
cube, ang = g["cube"], g["angles"]
for tag, kw in (("k4", dict(ncomp=4)), ("k9", dict(ncomp=9)), ("k4_mask", dict(ncomp=4, mask_center_px=6)), ("k9_mask", dict(ncomp=9, mask_center_px=6)),
                ("k12", dict(ncomp=12)), ("k9_mask_tm", dict(ncomp=9, mask_center_px=6, scaling="temp-mean"))):
    ref = O.pca_fullframe(cube, ang, **kw)                      # float64 oracle
    fr = pca(cube, ang, verbose=False, **kw)
    fr32 = pca(cube.astype(np.float32), ang, verbose=False, **kw)
    d = np.abs(fr - ref); d[np.isnan(d)] = 0
    iy, ix = np.unravel_index(d.argmax(), d.shape)
    print(tag, "device(f64 in) vs oracle(f64): %.3e at (%d,%d) r=%.1f  frame there %.3f / %.3f;  median |d| %.2e;  device(f32 in) vs oracle %.3e" % (
        d.max(), iy, ix, np.hypot(iy - 32, ix - 32), fr[iy, ix], ref[iy, ix], np.median(d), np.nanmax(np.abs(fr32 - ref))), flush=True)
    if tag in g.files or ("frame64_" + tag) in g.files:
        print("    oracle vs reference golden: %.3e" % np.nanmax(np.abs(ref - g["frame64_" + tag])))
