"""Final-frame deviation of the float64 route (csrc/pca_f64.hip) and of the float32 route from the float64 oracle on the g28 cube
(detector counts 7000 +- 45), and from the reference's own float64 goldens."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_cpu as O
from vip_amd.psfsub import pca
g = np.load("tests/golden/g28_f64_counts.npz")
cube, ang = g["cube"], g["angles"]
for tag, kw in (("k4", dict(ncomp=4)), ("k9", dict(ncomp=9)), ("k4_mask", dict(ncomp=4, mask_center_px=6)), ("k9_mask", dict(ncomp=9, mask_center_px=6)),
                ("k12", dict(ncomp=12)), ("k4_tm", dict(ncomp=4, scaling="temp-mean")), ("k9_mask_tm", dict(ncomp=9, mask_center_px=6, scaling="temp-mean")),
                ("k5_ts", dict(ncomp=5, scaling="temp-standard")), ("k4_mean", dict(ncomp=4, collapse="mean"))):
    ref = O.pca_fullframe(cube, ang, **kw)                      # float64 oracle
    fr = pca(cube, ang, verbose=False, **kw)
    fr32 = pca(cube.astype(np.float32), ang, verbose=False, **kw)
    d = np.abs(fr - ref); d[np.isnan(d)] = 0
    print("%-11s float64 route vs oracle(f64): %.3e (median %.1e);  float32 route: %.3e;  frame scale %.1f" % (
        tag, d.max(), np.median(d), np.nanmax(np.abs(fr32 - ref)), np.nanmax(np.abs(ref))), flush=True)
    if ("frame64_" + tag) in g.files:
        print("            vs the reference's float64 golden: %.3e" % np.nanmax(np.abs(fr - g["frame64_" + tag])))
