"""numpy-in / numpy-out latency of pca() at C2 and the raw host->device copy rates behind it."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
from vip_amd import backend as B
cube, ang = synth_adi(400, 512, 0)
ct = torch.from_numpy(cube).cuda(); torch.cuda.synchronize()
pca(ct, ang, ncomp=20, verbose=False, check_memory=False); torch.cuda.synchronize()
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
gb = cube.nbytes / 1e9
ms = t(lambda: torch.from_numpy(cube).cuda()); print("pageable H2D %.1f ms (%.1f GB/s)" % (ms, gb / ms * 1e3))
pin = torch.from_numpy(cube).pin_memory()
ms = t(lambda: pin.cuda(non_blocking=True)); print("pinned H2D %.1f ms (%.1f GB/s)" % (ms, gb / ms * 1e3))
ms = t(lambda: B.to_device_f32(cube)); print("to_device_f32 %.1f ms (%.1f GB/s)" % (ms, gb / ms * 1e3))
ms = t(lambda: pca(cube, ang, ncomp=20, verbose=False, check_memory=False)); print("pca(numpy cube) %.1f ms -> %.0f frames/s" % (ms, 400 / ms * 1e3))
ms = t(lambda: pca(ct, ang, ncomp=20, verbose=False, check_memory=False).cpu()); print("pca(device cube) %.1f ms" % ms)
c64 = cube.astype(np.float64)
ms = t(lambda: pca(c64, ang, ncomp=20, verbose=False, check_memory=False)); print("pca(float64 numpy cube) %.1f ms" % ms)
