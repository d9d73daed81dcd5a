import sys, os, time, warnings; sys.path.insert(0, ".")
import numpy as np, torch
warnings.simplefilter("ignore")
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca_annular
def t(fn, reps=3):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n, N in ((64, 101), (100, 128), (100, 201), (150, 256), (200, 301), (300, 401), (128, 512), (256, 256)):
    cube, _ = synth_adi(n, N, n); ct = torch.from_numpy(cube).cuda(); ang = np.sort(np.random.default_rng(n).uniform(0, 90, n))
    row = []
    for asz in (4, 16):
        for f in ("0", "1"):
            os.environ["VIPMI_ANNULAR_FUSED"] = f
            row.append(t(lambda: pca_annular(ct, ang, asize=asz, fwhm=4, ncomp=5, verbose=False)))
    print("%3d x %3d   asize 4: per-segment %7.2f  fused %7.2f ms     asize 16: per-segment %7.2f  fused %7.2f ms" % (n, N, *row), flush=True)
