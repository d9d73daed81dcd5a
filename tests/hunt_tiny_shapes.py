"""The smallest inputs: 1 .. 4 frames of 3 .. 14 px through pca / median_sub / cube_derotate / cube_collapse against the oracle --
same result, or the same kind of exception.   python tests/hunt_tiny_shapes.py"""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_cpu as O
from vip_amd.psfsub import pca, median_sub
from vip_amd.preproc import cube_derotate, cube_collapse
bad = 0
rng = np.random.default_rng(5)
for n in (1, 2, 3, 4):
    for N in (3, 4, 5, 6, 7, 9, 10, 13, 14):
        cube = rng.standard_normal((n, N, N)).astype(np.float32) * 3
        ang = np.linspace(0, 50, n)
        for name, dev, ora in (
            ("derotate", lambda: cube_derotate(cube, ang), lambda: O.cube_derotate(cube, ang)),
            ("collapse median", lambda: cube_collapse(cube, "median"), lambda: O.cube_collapse(cube, "median")),
            ("collapse trimmean", lambda: cube_collapse(cube, "trimmean", n=max(1, n - 1)), lambda: O.cube_collapse(cube, "trimmean", n=max(1, n - 1))),
            ("pca k1", lambda: pca(cube, ang, ncomp=1, verbose=False), lambda: O.pca_fullframe(cube, ang, ncomp=1)),
            ("pca k=n", lambda: pca(cube, ang, ncomp=n, verbose=False), lambda: O.pca_fullframe(cube, ang, ncomp=n)),
            ("pca k1 temp-standard", lambda: pca(cube, ang, ncomp=1, scaling="temp-standard", verbose=False), lambda: O.pca_fullframe(cube, ang, ncomp=1, scaling="temp-standard")),
            ("median_sub", lambda: median_sub(cube, ang, verbose=False), lambda: O.median_sub_fullfr(cube, ang)),
        ):
            try:
                want, werr = ora(), None
            except Exception as e:
                want, werr = None, e
            try:
                got, gerr = dev(), None
            except Exception as e:
                got, gerr = None, e
            if werr is not None or gerr is not None:
                if (werr is None) != (gerr is None):
                    bad += 1
                    print("FAIL n %d N %d %s: oracle %s, device %s" % (n, N, name, "ok" if werr is None else repr(werr)[:120], "ok" if gerr is None else repr(gerr)[:160]), flush=True)
                continue
            got, want = np.asarray(got), np.asarray(want)
            ok = np.isfinite(want)
            if got.shape != want.shape or not np.array_equal(np.isfinite(got), ok):
                bad += 1
                print("FAIL n %d N %d %s: shape / NaN pattern (%s vs %s)" % (n, N, name, got.shape, want.shape), flush=True)
                continue
            d = float(np.abs(got[ok] - want[ok]).max()) if ok.any() else 0.0
            # (a 1- or 2-frame PCA with k = n leaves residuals of pure rounding: compare on the scale of the data)
            if d > 1e-4 * max(1.0, float(np.abs(cube).max()) / 10):
                bad += 1
                print("FAIL n %d N %d %s: max|d| %.3e" % (n, N, name, d), flush=True)
print("failures:", bad)
