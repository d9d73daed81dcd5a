"""pca / pca_annular / median_sub are linear in the cube: pca(s * cube) / s against pca(cube) for s from 1e-25 to 1e25 (float32
input; the Gram matrix, the digit planes of the int8 product and the eigensolver must not overflow, underflow or lose digits),
plus cubes with a zero frame, duplicated frames and a constant offset of 1e6.   python tools/hunt_scale_invariance.py"""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vip_amd.psfsub import pca, pca_annular, median_sub
from vip_amd.synth import synth_adi
bad = 0
for (n, N) in ((40, 64), (400, 128), (90, 301)):
    cube, ang = synth_adi(n, N, seed=n)
    ang = np.linspace(0, 100, n)
    calls = {"pca k5": lambda c: pca(c, ang, ncomp=5, verbose=False),
             "pca k5 temp-standard": lambda c: pca(c, ang, ncomp=5, scaling="temp-standard", verbose=False),
             "pca k5 spat-mean mask": lambda c: pca(c, ang, ncomp=5, scaling="spat-mean", mask_center_px=5, verbose=False),
             "annular k3": lambda c: pca_annular(c, ang, ncomp=3, asize=8, fwhm=4, verbose=False),
             "median_sub": lambda c: median_sub(c, ang, verbose=False)}
    for name, fn in calls.items():
        base = fn(cube)
        sc = float(np.nanmax(np.abs(base)))
        for s in (1e-25, 1e-12, 1e-4, 1e5, 1e14, 1e25):
            if "standard" in name and s < 1e-4:
                continue        # (sklearn's scale treats a float32 column with sd < 10 eps = 1.2e-6 as constant: sd := 1 -- not scale-free)
            try:
                out = fn((cube * np.float32(s)).astype(np.float32))
                if "standard" not in name:
                    out = out / s
                ok = np.isfinite(base)
                assert np.array_equal(np.isfinite(out), ok), "NaN pattern"
                d = float(np.abs(out[ok] - base[ok]).max()) / sc
                assert d < 2e-4, "relative difference %.2e" % d
            except Exception as e:
                bad += 1
                print("FAIL %dx%dx%d %s s=%g: %s" % (n, N, N, name, s, "".join(traceback.format_exception_only(type(e), e)).strip()[:300]), flush=True)
    # special cubes
    for tag, c2 in (("zero frame", cube.copy()), ("duplicated frames", cube.copy()), ("offset 1e6", cube + np.float32(1e6))):
        if tag == "zero frame":
            c2[3] = 0
        if tag == "duplicated frames":
            c2[5] = c2[4]; c2[9] = c2[4]
        for name in ("pca k5", "pca k5 temp-standard", "annular k3", "median_sub"):
            try:
                out = calls[name](c2)
                assert np.isfinite(out[N // 2 - 3:N // 2 + 3, N // 2 + 8:N // 2 + 12]).all(), "non-finite result"
                if tag == "offset 1e6" and name in ("median_sub",):
                    ok = np.isfinite(base)
            except Exception as e:
                bad += 1
                print("FAIL %dx%dx%d %s on a cube with %s: %s" % (n, N, N, name, tag, "".join(traceback.format_exception_only(type(e), e)).strip()[:300]), flush=True)
    print("done %dx%dx%d" % (n, N, N), flush=True)
print("failures:", bad)
