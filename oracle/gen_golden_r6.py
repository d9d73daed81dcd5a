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
