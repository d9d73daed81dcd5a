"""Randomised differential runs at MEDIUM sizes (40..260 frames of 90..300 px, odd and even: the plans between the fuzz suite's
small cubes and the BASELINE shapes -- wave-resident / multi-workgroup eigensolvers, LDS subtraction tiles, the power-of-two and the
circular-convolution shears): pca / pca_annular / float64 routes against the oracle.   python tools/fuzz_medium.py [first [count]]"""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_cpu as O
from vip_amd.psfsub import pca, pca_annular, median_sub

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
TOL = 1e-4
SCALINGS = (None, "temp-mean", "spat-mean", "temp-standard", "spat-standard")
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(40, 260))
    N = int(rng.integers(90, 300))
    cube, _ = O.synth_adi(n, N, seed=int(rng.integers(1 << 30)))
    cube = cube.astype(np.float32)
    ang = np.linspace(0, float(rng.uniform(40, 200)), n) if rng.integers(2) else np.sort(rng.uniform(-150, 150, n))
    scaling = SCALINGS[rng.integers(len(SCALINGS))]
    kind = seed % 4
    t0 = time.time()
    try:
        if kind == 0:
            kw = dict(ncomp=int(rng.integers(1, min(n, 40))), scaling=scaling, collapse=("median", "mean", "trimmean")[rng.integers(3)])
            if rng.integers(3) == 0:
                kw["mask_center_px"] = int(rng.integers(3, N // 6))
            ref = O.pca_fullframe(cube, ang, **kw); out = pca(cube, ang, verbose=False, **kw)
        elif kind == 1:
            kw = dict(ncomp=int(rng.integers(1, 12)), scaling=scaling, asize=int(rng.integers(6, 20)), fwhm=4, delta_rot=(0.1, float(rng.uniform(0.4, 1.0))),
                      n_segments=int(rng.integers(1, 3)), radius_int=int(rng.integers(0, 8)))
            ang = np.linspace(0, float(rng.uniform(60, 200)), n)
            ref = O.pca_annular(cube, ang, **kw); out = pca_annular(cube, ang, verbose=False, **kw)
        elif kind == 2:
            c64 = 7000.0 + 45.0 * cube.astype(np.float64)
            kw = dict(ncomp=int(rng.integers(1, 20)), scaling=scaling)
            ref = O.pca_fullframe(c64, ang, **kw); out = pca(c64, ang, verbose=False, **kw)
        else:
            kw = dict(mode="annular", asize=int(rng.integers(4, 10)), fwhm=4, delta_rot=float(rng.uniform(0.3, 1.0)), nframes=int(rng.integers(2, 6)) * 2)
            ang = np.linspace(0, float(rng.uniform(60, 200)), n)
            ref = O.median_sub_annular(cube, ang, **{a: b for a, b in kw.items() if a != "mode"}); out = median_sub(cube, ang, verbose=False, **kw)
        ok = np.isfinite(ref)
        assert out.shape == ref.shape and np.array_equal(np.isfinite(out), ok), "shape / NaN pattern"
        tol = TOL * max(1.0, float(np.abs(ref[ok]).max()) / 10.0)
        d = float(np.abs(out[ok] - ref[ok]).max())
        assert d < tol, "max|d| %.3e >= %.3e" % (d, tol)
        print("ok   seed %d kind %d n %d N %d %s: %.2e  (%.1f s)" % (seed, kind, n, N, kw, d, time.time() - t0), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d kind %d n %d N %d %s: %s" % (seed, kind, n, N, kw, "".join(traceback.format_exception_only(type(e), e)).strip()[:500]), flush=True)
print("failures:", bad)
