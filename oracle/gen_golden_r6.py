"""Fixtures added in round 6 (G29 ...): outputs of the REAL reference (imported read-only through oracle/_shim.py) frozen
as data under tests/golden/; runs only in the build container:

    python oracle/gen_golden_r6.py [g29 ...]
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


# ---- G29: median_sub with an ODD frame count and radius_int > 0 (round-5 ADVICE).  np.median of an odd number of samples IS one
# of the samples, so `cube - median` is exactly 0 for one frame per pixel; with radius_int the reference rotates with mask_val = 0
# (psfsub/medsub.py:262-266), which treats those zeros as masked pixels and resets them after the rotation
# (preproc/derotation.py:133-140,324-326).  Both modes, float32 cube (the reference then subtracts in float32: the zeros are exact).
if want("g29"):
    n, N = 17, 48
    cube, _ = O.synth_adi(n, N, seed=2900)
    ang = np.linspace(0, 70, n)
    g = {"cube": cube, "angles": ang}
    for tag, kw in (("ff", dict(radius_int=5)), ("ff_mean", dict(radius_int=3, collapse="mean")),
                    ("ann", dict(mode="annular", asize=4, fwhm=4, radius_int=4, nframes=4))):
        co, cd, fr_ = ref.median_sub(cube, ang, full_output=True, verbose=False, nproc=1, **kw)
        g["ms_%s_out" % tag], g["ms_%s_der" % tag], g["ms_%s_frame" % tag] = co, cd, fr_
        print("   %s: exact zeros in cube_out outside the mask: %d" % (tag, int((np.asarray(co) == 0).sum())))
    save("g29_medsub_odd", **g)


# ---- G30: the SPATIAL scalings on a float64 cube of detector counts (g28's cube, not stored again): the reference's own float64
# frames and residual cubes for matrix_scaling(axis=1) (var/shapes.py:740-781), and -- the yardstick -- its result when the cube is
# handed over as float32.
if want("g30"):
    g28 = np.load(os.path.join(OUT, "g28_f64_counts.npz"))
    cube, ang = g28["cube"], g28["angles"]
    g = {}
    for tag, kw in (("smean", dict(ncomp=4, scaling="spat-mean")), ("sstd", dict(ncomp=5, scaling="spat-standard")),
                    ("sstd_mask", dict(ncomp=3, scaling="spat-standard", mask_center_px=6))):
        fo = ref.pca(cube, ang, full_output=True, verbose=False, nproc=1, **kw)
        g["frame64_" + tag] = np.asarray(fo[0], dtype=np.float64)
        if tag == "smean":
            g["res64_" + tag] = np.asarray(fo[3], dtype=np.float32)
        f32 = ref.pca(cube.astype(np.float32), ang, full_output=False, verbose=False, nproc=1, **kw)
        g["frame_ref_f32_" + tag] = np.asarray(f32, dtype=np.float64)
        d = np.nanmax(np.abs(g["frame_ref_f32_" + tag] - g["frame64_" + tag]))
        o = O.pca_fullframe(cube, ang, **kw)
        print("   %s: reference(float32 cube) vs reference(float64 cube): max|d| = %.3e; oracle vs reference %.3e (frame scale %.3f)"
              % (tag, d, np.nanmax(np.abs(o - g["frame64_" + tag])), np.nanmax(np.abs(g["frame64_" + tag]))))
    save("g30_f64_spat", **g)
