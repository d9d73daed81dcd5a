"""Time the interpolating rotation (imlib='opencv') and pca() with it: python tools/time_warp.py"""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
n, N = 400, 512
cube = torch.randn(n, N, N, device="cuda"); ang = np.linspace(0, 90, n)
out = torch.empty_like(cube)
def tm(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for it in ("nearneig", "bilinear", "bicubic", "lanczos4"):
    ms = tm(lambda: B.rotate_interp(cube, ang, it, out=out))
    print("%-9s %.3f ms per %d x %d^2 (%.0f GB/s of the 8 B / pixel read + write)" % (it, ms, n, N, 8.0 * n * N * N / ms / 1e6))
print("vip-fft   %.3f ms" % tm(lambda: B.derotate(cube, ang, out=out)))
for lib, it in (("vip-fft", "lanczos4"), ("opencv", "lanczos4"), ("opencv", "bilinear")):
    print("pca(ncomp=20, imlib=%r, %r): %.3f ms" % (lib, it, tm(lambda: pca(cube, ang, ncomp=20, imlib=lib, interpolation=it, verbose=False), 5)))
