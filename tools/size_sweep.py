"""pca / pca_annular / median_sub over cube sizes (numpy in, numpy out, as a VIP caller passes them): first call (plans, tables) and
steady state."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
warnings.simplefilter("ignore")
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca, pca_annular, median_sub
def t(fn, reps=3):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return first, (time.perf_counter() - t0) / reps * 1e3
pca(synth_adi(20, 64, 1)[0], np.linspace(0, 50, 20), ncomp=2, verbose=False)
print("%-14s %22s %26s %26s %22s" % ("n x N", "pca k=10", "pca_annular asize=4 k=5", "pca_annular asize=16 k=5", "median_sub"))
for n, N in ((50, 64), (61, 101), (100, 128), (100, 201), (150, 256), (200, 301), (300, 401), (400, 512)):
    cube, _ = synth_adi(n, N, n); ang = np.sort(np.random.default_rng(n).uniform(0, 90, n))      # (real PA lists have no ties)
    a = t(lambda: pca(cube, ang, ncomp=10, verbose=False))
    b = t(lambda: pca_annular(cube, ang, asize=4, fwhm=4, ncomp=5, verbose=False), 2)
    c = t(lambda: pca_annular(cube, ang, asize=16, fwhm=4, ncomp=5, verbose=False), 2)
    d = t(lambda: median_sub(cube, ang, verbose=False))
    print("%-14s %9.2f /%9.2f ms %12.2f /%9.2f ms %12.2f /%9.2f ms %9.2f /%9.2f ms" % ("%d x %d" % (n, N), *a, *b, *c, *d), flush=True)
