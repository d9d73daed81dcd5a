import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca_annular
cubes = np.stack([synth_adi(100, 128, s)[0] for s in range(10)]); ang = np.linspace(0, 90, 100)
ct = torch.from_numpy(cubes).cuda()
f = lambda: pca_annular(ct, ang, asize=8, ncomp=5, fwhm=4, delta_rot=(0.1, 1), verbose=False).cpu()
f(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): f()
torch.cuda.synchronize(); print("annular 4-D 10x100x128x128: %.2f ms" % ((time.perf_counter() - t) / 3 * 1e3))
