"""cube_rescaling_wavelengths (the FFT zoom of ADI+mSDI, preproc/rescaling.py:427-475 with imlib='vip-fft') at random frame sizes
(odd and even, 12 .. 320 px), channel counts and scale lists, forward and inverse with the crop back -- against the oracle.
   python tests/hunt_rescaling_sizes.py [first [count]]"""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_cpu as O
from vip_amd.preproc.rescaling import cube_rescaling_wavelengths

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bad = 0
t00 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(19000 + seed)
    N = int(rng.integers(12, 320)); nch = int(rng.integers(2, 9))
    top = float(rng.uniform(1.02, 1.7))
    scal = np.linspace(top, 1.0, nch) if rng.integers(2) else np.sort(rng.uniform(1.0, top, nch))[::-1].copy()
    if rng.integers(4) == 0:
        scal = scal / scal[0]                       # all <= 1: down-scaling, no padding
    cube = (rng.standard_normal((nch, N, N)) * rng.uniform(0.5, 5)).astype(np.float32)
    yy, xx = np.mgrid[:N, :N]
    cube += (20 * np.exp(-((yy - N // 2) ** 2 + (xx - N // 2) ** 2) / (2 * (N / 12.0) ** 2))).astype(np.float32)
    coll = ("median", "mean")[rng.integers(2)]
    what = "N %d nch %d scales %.3f..%.3f %s" % (N, nch, scal.max(), scal.min(), coll)
    try:
        ref = O.cube_rescaling_wavelengths(cube, scal, full_output=True, collapse=coll)
        got = cube_rescaling_wavelengths(cube, scal, full_output=True, collapse=coll)
        tol = 5e-5 * max(1.0, float(np.abs(cube).max()) / 10)
        for i, nm in ((0, "cube"), (1, "frame")):
            a, b = np.asarray(got[i]), np.asarray(ref[i])
            assert a.shape == b.shape, "%s shape %s vs %s" % (nm, a.shape, b.shape)
            d = float(np.nanmax(np.abs(a - b)))
            assert d < tol, "forward %s: max|d| %.2e" % (nm, d)
        assert tuple(got[2:]) == tuple(ref[2:]), "geometry %s vs %s" % (got[2:], ref[2:])
        if scal.max() > 1:
            big = np.asarray(ref[0]).astype(np.float32)
            ri = O.cube_rescaling_wavelengths(big, scal, full_output=True, inverse=True, y_in=N, x_in=N, collapse=coll)
            gi = cube_rescaling_wavelengths(big, scal, full_output=True, inverse=True, y_in=N, x_in=N, collapse=coll)
            for i, nm in ((0, "cube"), (1, "frame")):
                a, b = np.asarray(gi[i]), np.asarray(ri[i])
                assert a.shape == b.shape, "inverse %s shape %s vs %s" % (nm, a.shape, b.shape)
                d2 = float(np.nanmax(np.abs(a - b)))
                assert d2 < tol, "inverse %s: max|d| %.2e" % (nm, d2)
        print("ok   seed %d %s: %.2e  (%.0f s so far)" % (seed, what, d, time.time() - t00), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d %s: %s" % (seed, what, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
print("failures:", bad)
