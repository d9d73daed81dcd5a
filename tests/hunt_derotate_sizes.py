"""cube_derotate at random frame sizes 12 .. 1100 px (every plan family: power-of-two padded periods, the real-split direct path,
its circular-convolution passes, the two Le = 4096 plans) against the oracle's frame_rotate_fft -- random angles in every quadrant,
a NaN patch, mask_val 0 and NaN; default method and method='direct'.   python tests/hunt_derotate_sizes.py [first [count]]"""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_cpu as O
from vip_amd.preproc import cube_derotate

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
t00 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(11000 + seed)
    N = int(rng.integers(12, 1100)) if seed % 3 else int(rng.choice([64, 128, 256, 512, 1024, 96, 160, 320, 640, 1000, 513, 1023, 1025, 257, 129, 127]))
    n = int(rng.integers(2, 6)) if N < 600 else 2
    cube = (rng.standard_normal((n, N, N)) * rng.uniform(0.5, 5)).astype(np.float32)
    ang = rng.uniform(-400, 400, n)
    ang[0] = float(rng.choice([0.0, 45.0, -45.0, 90.0, 135.0, 180.0, 44.999, 45.001, 359.99]))
    if seed % 2:
        cube[n - 1, N // 3:N // 3 + 3, N // 4] = np.nan
    try:
        ref = O.cube_derotate(cube, ang)
        for method in ("auto", "direct"):
            got = cube_derotate(cube, ang, method=method) if method != "auto" else cube_derotate(cube, ang)
            assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref)), "NaN pattern (%s)" % method
            d = float(np.nanmax(np.abs(got - ref)))
            assert d < 5e-5 * max(1.0, float(np.nanmax(np.abs(cube))) / 10.0), "%s: max|d| %.3e" % (method, d)
        print("ok   seed %d N %d n %d: %.2e  (%.0f s so far)" % (seed, N, n, d, time.time() - t00), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d N %d n %d angles %s: %s" % (seed, N, n, np.round(ang, 3), "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
print("failures:", bad)
