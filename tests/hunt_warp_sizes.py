"""cube_derotate(imlib='opencv') -- the interpolating rotation (csrc/warp.hip) -- at random frame sizes, interpolations, border
modes and rotation centres against the oracle's warp_rotate (a restatement of the published algorithm: parity unpinned, no cv2
here); cube_derotate rotates frame i by -angle_list[i] (derotation.py:395).   python tests/hunt_warp_sizes.py [first [count]]"""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_cpu as O
from vip_amd.preproc import cube_derotate

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bad = 0
t00 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(29000 + seed)
    N = int(rng.integers(8, 400)); n = int(rng.integers(1, 5))
    cube = (rng.standard_normal((n, N, N)) * 3).astype(np.float32)
    ang = rng.uniform(-360, 360, n)
    interp = ("nearneig", "bilinear", "bicubic", "lanczos4")[rng.integers(4)]
    border = ("constant", "edge", "symmetric", "reflect", "wrap")[rng.integers(5)]
    cxy = None if rng.integers(2) else (float(rng.uniform(N / 3, 2 * N / 3)), float(rng.uniform(N / 3, 2 * N / 3)))
    what = "N %d n %d %s %s cxy %s" % (N, n, interp, border, None if cxy is None else "(%.2f, %.2f)" % cxy)
    try:
        ref = np.stack([O.warp_rotate(cube[i], -ang[i], interpolation=interp, cxy=cxy, border_mode=border) for i in range(n)])
        got = cube_derotate(cube, ang, imlib="opencv", interpolation=interp, cxy=cxy, border_mode=border)
        assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref)), "shape / NaN pattern"
        d = np.abs(got - ref)
        d[~np.isfinite(d)] = 0
        if interp == "nearneig":                 # a tie in the rounding of a source coordinate may pick the neighbour: a handful of pixels
            assert (d > 1e-4).sum() <= max(3, d.size // 2000), "nearest neighbour: %d pixels differ" % int((d > 1e-4).sum())
        else:
            assert d.max() < 2e-4, "max|d| %.2e" % d.max()
        print("ok   seed %d %s: %.2e  (%.0f s so far)" % (seed, what, d.max(), time.time() - t00), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d %s: %s" % (seed, what, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
print("failures:", bad)
