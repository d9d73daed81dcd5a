"""Times the BASELINE.json configs that fit one GPU (C1..C4) through the public API."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca, pca_annular

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

which = sys.argv[1:] or ["c1", "c2", "c3", "c4"]
if "c1" in which:
    cube, ang = synth_adi(50, 128, 0); ct = torch.from_numpy(cube).cuda()
    t = timeit(lambda: pca(ct, ang, ncomp=5, verbose=False, check_memory=False).cpu(), 10)
    print("C1 50x128x128 k=5: %.3f ms  %.0f frames/s" % (t * 1e3, 50 / t))
if "c2" in which:
    cube, ang = synth_adi(400, 512, 0); ct = torch.from_numpy(cube).cuda()
    t = timeit(lambda: pca(ct, ang, ncomp=20, verbose=False, check_memory=False).cpu(), 5)
    print("C2 400x512x512 k=20: %.3f ms  %.0f frames/s" % (t * 1e3, 400 / t))
if "c3" in which:
    cube, ang = synth_adi(400, 512, 0); ct = torch.from_numpy(cube).cuda()
    t = timeit(lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, delta_rot=(0.1, 1), verbose=False).cpu(), 2)
    print("C3 400x512x512 annular 8 annuli k=10: %.3f ms  %.0f frames/s" % (t * 1e3, 400 / t))
if "c4" in which:
    cubes = np.stack([synth_adi(200, 256, s)[0] for s in range(39)]); ang = np.linspace(0, 90, 200)
    ct = torch.from_numpy(cubes).cuda()
    t = timeit(lambda: pca(ct, ang, ncomp=20, verbose=False, check_memory=False).cpu(), 2)
    print("C4 39x200x256x256 k=20: %.3f ms  %.0f frames/s" % (t * 1e3, 39 * 200 / t))
if "odd" in which:        # C2 with the reference's odd frame size: Le = 2044 is not a power of two (csrc/derotate_conv.inc)
    cube, ang = synth_adi(400, 511, 0); ct = torch.from_numpy(cube).cuda()
    t = timeit(lambda: pca(ct, ang, ncomp=20, verbose=False, check_memory=False).cpu(), 5)
    print("400x511x511 k=20: %.3f ms  %.0f frames/s" % (t * 1e3, 400 / t))
