import sys, time, cProfile, pstats; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca, pca_annular, median_sub
cube, ang = synth_adi(400, 512, 0); ct = torch.from_numpy(cube).cuda()
ref, _ = synth_adi(100, 512, 5); rt = torch.from_numpy(ref).cuda()
for nm, fn in (("median_sub annular", lambda: median_sub(ct, ang, mode="annular", asize=32, fwhm=4, delta_rot=1, nframes=4, verbose=False)),
               ("pca source_xy", lambda: pca(ct, ang, ncomp=10, source_xy=(300, 256), fwhm=4, delta_rot=1, verbose=False)),
               ("pca_annular first call (plan not cached) asize=16", lambda: pca_annular(ct, ang, asize=16, ncomp=7, fwhm=4, verbose=False))):
    if "first" not in nm: fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0; pr.disable()
    print("==", nm, "%.1f ms" % (dt * 1e3)); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
