"""Fixtures added in round 5 (G28 ...): outputs of the REAL reference (imported read-only through oracle/_shim.py) frozen
as data under tests/golden/; runs only in the build container:

    python oracle/gen_golden_r5.py [g28 ...]
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _shim, ref_cpu as O  # noqa: E402

warnings.simplefilter("ignore")
ref = _shim.load()
OUT = os.path.join(ROOT, "tests", "golden")
WHICH = set(sys.argv[1:])


def want(name):
    return not WHICH or name in WHICH


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# ---- G28: a float64 cube with detector-count values (~7e3, SURVEY 7 "fp32 Gram conditioning").  The reference keeps the
# input dtype through svd_wrapper (psfsub/pca_fullfr.py:1552-1737, SURVEY a9 "f64 if f64 in"); the device path converts to
# float32 on upload.  Frozen: the reference's float64 result, and -- as the yardstick -- the reference's OWN result when the
# same cube is handed over as float32 (its LAPACK / FFT then run in float32).
if want("g28"):
    n, N = 36, 64
    base, ang = O.synth_adi(n, N, seed=2800)
    rng = np.random.default_rng(2801)
    cube = 7000.0 + 45.0 * base.astype(np.float64) + 1e-4 * rng.standard_normal(base.shape)      # not float32-representable
    g = {"cube": cube, "angles": ang}
    for tag, kw in (("k4", dict(ncomp=4)), ("k4_tm", dict(ncomp=4, scaling="temp-mean")), ("k9_mask", dict(ncomp=9, mask_center_px=6))):
        fo = ref.pca(cube, ang, full_output=True, verbose=False, nproc=1, **kw)
        g["frame64_" + tag] = np.asarray(fo[0], dtype=np.float64)
        g["res64_" + tag] = np.asarray(fo[3], dtype=np.float64)
        f32 = ref.pca(cube.astype(np.float32), ang, full_output=False, verbose=False, nproc=1, **kw)
        g["frame_ref_f32_" + tag] = np.asarray(f32, dtype=np.float64)
        d = np.nanmax(np.abs(g["frame_ref_f32_" + tag] - g["frame64_" + tag]))
        print("   %s: reference(float32 cube) vs reference(float64 cube): max|d| = %.3e   (max|cube| = %.1f, frame scale %.2f)"
              % (tag, d, np.abs(cube).max(), np.nanmax(np.abs(g["frame64_" + tag]))))
    save("g28_f64_counts", **g)
