"""pca_annular / pca_annulus / pca where the matrix has FEWER pixels than frames (thin annuli, many segments, tiny frames): the
library Gram matrices are rank-deficient, ncomp is clipped to the number of pixels (svd.py:696) -- against the oracle.
   python tests/hunt_rank_deficient_segments.py"""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_cpu as O
from vip_amd.psfsub import pca, pca_annular
bad = 0
for seed in range(12):
    rng = np.random.default_rng(47000 + seed)
    n = int(rng.integers(60, 140)); N = int(rng.integers(30, 52))
    cube = O.synth_adi(n, N, seed=seed)[0].astype(np.float32)
    ang = np.linspace(0, float(rng.uniform(100, 300)), n)
    kw = dict(ncomp=int(rng.integers(2, 30)), asize=int(rng.integers(2, 4)), fwhm=int(rng.integers(2, 4)), n_segments=int(rng.choice([1, 2, 4, 6])),
              delta_rot=(0.1, float(rng.uniform(0.3, 1))), radius_int=int(rng.integers(0, 4)), scaling=(None, "temp-mean", "temp-standard")[rng.integers(3)],
              min_frames_lib=int(rng.integers(2, 10)))
    t0 = time.time()
    try:
        ref = O.pca_annular(cube, ang, **kw)
        out = pca_annular(cube, ang, verbose=False, **kw)
        ok = np.isfinite(ref)
        assert out.shape == ref.shape and np.array_equal(np.isfinite(out), ok), "shape / NaN pattern"
        d = float(np.abs(out[ok] - ref[ok]).max())
        if d >= 1e-4 * max(1.0, float(np.abs(ref[ok]).max()) / 10) and kw["radius_int"] > 0:
            # ncomp >= the pixels of a segment: its residuals are pure cancellation, a quarter of them exactly 0.0f in the oracle's
            # float32 arithmetic, and with radius_int > 0 the reference rotates with mask_val = 0 -- every exact zero a masked pixel
            # (derotation.py:133-140): the frame depends on the last bit of those cancellations (NOTES round 6).  The residual cubes agree.
            ro = O.pca_annular(cube, ang, full_output=True, **kw)
            go = pca_annular(cube, ang, verbose=False, full_output=True, **kw)
            dres = float(np.abs(np.asarray(go[0]) - np.asarray(ro[0])).max())
            zeros = int((np.asarray(ro[0]) == 0).sum())
            assert dres < 5e-5, "residual cubes differ: %.2e" % dres
            print("note seed %d n %d N %d %s: frame %.2e apart, residual cubes %.2e apart, %d exact zeros in the oracle's residual cube" % (seed, n, N, kw, d, dres, zeros), flush=True)
            continue
        assert d < 1e-4 * max(1.0, float(np.abs(ref[ok]).max()) / 10), "max|d| %.2e" % d
        print("ok   seed %d n %d N %d %s: %.2e (%.0f s)" % (seed, n, N, kw, d, time.time() - t0), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d n %d N %d %s: %s" % (seed, n, N, kw, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
# full-frame: more frames than pixels
for seed in range(8):
    rng = np.random.default_rng(48000 + seed)
    n = int(rng.integers(40, 300)); N = int(rng.integers(4, 12))
    cube = (rng.standard_normal((n, N, N)) * 3).astype(np.float32)
    ang = np.linspace(0, 90, n)
    k = int(rng.integers(1, N * N + 1))
    for svd_mode in ("lapack", "eigen"):
        try:
            ref = O.pca_fullframe(cube, ang, ncomp=min(k, n), svd_mode=svd_mode)
            out = pca(cube, ang, ncomp=min(k, n), svd_mode=svd_mode, verbose=False)
            ok = np.isfinite(ref)
            d = float(np.abs(out[ok] - ref[ok]).max())
            assert np.array_equal(np.isfinite(out), ok) and d < 2e-4, "max|d| %.2e" % d
            print("ok   tall seed %d n %d N %d k %d %s: %.2e" % (seed, n, N, k, svd_mode, d), flush=True)
        except Exception as e:
            bad += 1
            print("FAIL tall seed %d n %d N %d k %d %s: %s" % (seed, n, N, k, svd_mode, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
print("failures:", bad)
