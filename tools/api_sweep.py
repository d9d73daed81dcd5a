"""Wall time of the public entry points at C2 size (400 x 512 x 512, resident float32 cube unless noted): where does a drop-in
caller land on a slow path?   python tools/api_sweep.py"""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
warnings.simplefilter("ignore")
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca, pca_annular, median_sub
from vip_amd.psfsub.svd import svd_wrapper
from vip_amd.psfsub.utils_pca import pca_grid
from vip_amd.preproc import cube_derotate, cube_collapse, frame_rotate
n, N = 400, 512
cube, ang = synth_adi(n, N, 0); ct = torch.from_numpy(cube).cuda()
ref, _ = synth_adi(100, N, 5); rt = torch.from_numpy(ref).cuda()
def t(name, fn, reps=3):
    try:
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); print("%-58s %8.2f ms" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    except Exception as e:
        print("%-58s %s: %s" % (name, type(e).__name__, str(e)[:90]), flush=True)
kw = dict(verbose=False)
t("pca k=20", lambda: pca(ct, ang, ncomp=20, **kw))
t("pca k=20 temp-mean", lambda: pca(ct, ang, ncomp=20, scaling="temp-mean", **kw))
t("pca k=20 temp-standard", lambda: pca(ct, ang, ncomp=20, scaling="temp-standard", **kw))
t("pca k=20 spat-standard", lambda: pca(ct, ang, ncomp=20, scaling="spat-standard", **kw))
t("pca k=20 mask_center_px=10", lambda: pca(ct, ang, ncomp=20, mask_center_px=10, **kw))
t("pca k=20 collapse=mean", lambda: pca(ct, ang, ncomp=20, collapse="mean", **kw))
t("pca k=20 collapse=trimmean", lambda: pca(ct, ang, ncomp=20, collapse="trimmean", **kw))
t("pca k=20 full_output (device)", lambda: pca(ct, ang, ncomp=20, full_output=True, **kw))
t("pca k=100", lambda: pca(ct, ang, ncomp=100, **kw))
t("pca ncomp=0.9 (CEVR)", lambda: pca(ct, ang, ncomp=0.9, **kw))
t("pca RDI cube_ref 100 frames k=20", lambda: pca(ct, ang, ncomp=20, cube_ref=rt, **kw))
t("pca ARDI", lambda: pca(ct, ang, ncomp=20, cube_ref=rt, ref_strategy="ARDI", **kw))
t("pca grid ncomp=(1,21,4) tuple", lambda: pca(ct, ang, ncomp=(1, 21, 4), **kw), 2)
t("pca list ncomp=[5,10,20]", lambda: pca(ct, ang, ncomp=[5, 10, 20], **kw), 2)
t("pca source_xy frame rejection k=10", lambda: pca(ct, ang, ncomp=10, source_xy=(300, 256), fwhm=4, delta_rot=1, **kw), 1)
t("pca imlib=opencv", lambda: pca(ct, ang, ncomp=20, imlib="opencv", **kw))
t("pca float64 resident", lambda c=ct.double(): pca(c, ang, ncomp=20, **kw))
t("pca_annular C3", lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, **kw))
t("pca_annular asize=16 (16 annuli)", lambda: pca_annular(ct, ang, asize=16, ncomp=10, fwhm=4, **kw), 2)
t("pca_annular asize=4 fwhm=4 (64 annuli, the defaults' scale)", lambda: pca_annular(ct, ang, asize=4, ncomp=5, fwhm=4, **kw), 1)
t("pca_annular n_segments=4", lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, n_segments=4, **kw), 2)
t("pca_annular ncomp tuple, radius_int=16", lambda: pca_annular(ct, ang, asize=32, ncomp=(4, 5, 6, 7, 8, 9, 10), fwhm=4, radius_int=16, **kw), 2)
t("pca_annular temp-mean", lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, scaling="temp-mean", **kw), 2)
t("pca_annular spat-mean", lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, scaling="spat-mean", **kw), 2)
t("pca_annular list ncomp [5,10]", lambda: pca_annular(ct, ang, asize=32, ncomp=[5, 10], fwhm=4, **kw), 2)
t("pca_annular cube_ref 100", lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, cube_ref=rt, **kw), 2)
t("pca_annular full_output (device)", lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, full_output=True, **kw), 2)
t("median_sub fullfr", lambda: median_sub(ct, ang, **kw))
t("median_sub annular asize=32", lambda: median_sub(ct, ang, mode="annular", asize=32, fwhm=4, delta_rot=1, nframes=4, **kw), 2)
t("cube_derotate", lambda: cube_derotate(ct, ang))
t("cube_derotate opencv", lambda: cube_derotate(ct, ang, imlib="opencv"))
for m in ("median", "mean", "sum", "trimmean", "max", "absmean"):
    t("cube_collapse " + m, lambda m=m: cube_collapse(ct, m))
t("frame_rotate one frame", lambda: frame_rotate(ct[0], 33.0), 20)
M = ct.reshape(n, -1)
for mode in ("lapack", "eigen", "randsvd", "arpack"):
    t("svd_wrapper %s k=20" % mode, lambda mode=mode: svd_wrapper(M, mode, 20, False, to_numpy=False))
t("svd_wrapper lapack full_output", lambda: svd_wrapper(M, "lapack", 20, False, full_output=True, to_numpy=False))
t("pca_grid range 1..20 step 4", lambda: pca_grid(ct, ang, range_pcs=(1, 21, 4), verbose=False, plot=False, full_output=False), 1)
