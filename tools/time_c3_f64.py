"""C3 with a float64 cube (numpy's default dtype): the float64 route of pca_annular against the float32 call."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca_annular
cube, ang = synth_adi(400, 512, 0)
c32 = torch.from_numpy(cube).cuda(); c64 = c32.double()
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
kw = dict(asize=32, ncomp=10, fwhm=4, delta_rot=(0.1, 1), verbose=False)
print("C3 float32 resident: %.2f ms" % t(lambda: pca_annular(c32, ang, **kw)))
print("C3 float64 resident: %.2f ms" % t(lambda: pca_annular(c64, ang, **kw)))
a = pca_annular(c32, ang, **kw).cpu().numpy(); b = pca_annular(c64, ang, **kw).cpu().numpy()
print("max |frame64 - frame32| = %.2e" % np.nanmax(np.abs(a - b)))
