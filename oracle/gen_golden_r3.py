"""Fixtures added in round 3 (G25 ...): outputs of the REAL reference (imported read-only through oracle/_shim.py) frozen
as data under tests/golden/; runs only in the build container:

    python oracle/gen_golden_r3.py [g25 g26 ...]
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _shim, ref_cpu as O  # noqa: E402

warnings.simplefilter("ignore")
ref = _shim.load()
OUT = os.path.join(ROOT, "tests", "golden")
WHICH = set(sys.argv[1:])


def want(name):
    return not WHICH or name in WHICH


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# ---- G25: frame_rotate / cube_derotate / pca with a `mask_val` that is neither NaN nor 0 (preproc/derotation.py:133-140,
# 324-326): pixels equal to mask_val are rotated WITH their value and reset to mask_val afterwards, NaN pixels are
# rotated as 0 and not restored.  Frame sizes on the FFT path (64 -> padded length 256 is not a device plan: the
# correlations; 128 -> 512: the wave-resident transforms) and an odd one.
if want("g25"):
    g = {}
    rng = np.random.default_rng(2500)
    for N in (40, 128, 45):
        fr = (rng.standard_normal((3, N, N)) * 2).astype(np.float32)
        yy, xx = np.mgrid[:N, :N]
        fr[:, np.hypot(yy - N / 2, xx - N / 2) > 0.47 * N] = 5.0          # a ring of "masked" pixels at the value 5
        fr[1, 3:6, 4:9] = np.nan                                            # NaNs that the mask value does not cover
        fr[2, N // 2, N // 2 - 3:N // 2 + 3] = -1.5
        angs = np.array([-17.3, 100.0, 211.0])
        g["in_%d" % N] = fr
        g["angles_%d" % N] = angs
        g["out5_%d" % N] = np.asarray(ref.cube_derotate(fr, angs, imlib="vip-fft", mask_val=5.0), dtype=np.float32)
        g["outm_%d" % N] = np.asarray(ref.cube_derotate(fr, angs, imlib="vip-fft", mask_val=-1.5), dtype=np.float32)
        g["fr5_%d" % N] = np.asarray(ref.frame_rotate(fr[0], 33.0, imlib="vip-fft", mask_val=5.0))
    # a value float32 cannot hold never matches a float32 pixel (the comparison is made in float64)
    fr = np.full((2, 40, 40), np.float32(0.1), dtype=np.float32)
    fr[:, 10:30, 10:30] = rng.standard_normal((2, 20, 20)).astype(np.float32)
    g["in_01"] = fr
    g["out_01"] = np.asarray(ref.cube_derotate(fr, np.array([20.0, -50.0]), imlib="vip-fft", mask_val=0.1), dtype=np.float32)
    # pca() with rot_options: mask_center_px + mask_val = 2.5 (the masked disk is 0 after mask_circle, so the value only
    # matters where residuals happen to equal it: nowhere -- the frames must equal the unmasked-rotation ones), and a cube
    # whose corners carry the value
    cube, ang = O.synth_adi(14, 40, seed=250)
    cube = cube.astype(np.float32)
    g["cube"], g["angles"] = cube, ang
    fo = ref.pca(cube, ang, ncomp=3, mask_center_px=5, mask_val=2.5, full_output=True, verbose=False, nproc=1)
    g["pca_frame"], g["pca_resder"] = np.asarray(fo[0]), np.asarray(fo[4])
    g["pca_nomask"] = np.asarray(ref.pca(cube, ang, ncomp=2, mask_val=7.0, verbose=False, nproc=1))
    from vip_hci.psfsub import median_sub
    g["medsub"] = np.asarray(median_sub(cube, ang, mask_val=1e3, verbose=False, nproc=1))
    save("g25_mask_val", **g)


# ---- G26: ADI+mSDI double pass with a reference cube and / or a rotation threshold at source_xy
# (psfsub/pca_fullfr.py:1279-1283,1388-1400,1403-1459) -------------------------------------------------------------
if want("g26"):
    z_, n_, N_ = 4, 12, 32
    c4 = np.stack([O.synth_adi(n_, N_, seed=180 + i)[0] for i in range(z_)]).astype(np.float32)
    a4 = np.linspace(0, 85, n_)
    sc = np.linspace(1.0, 1.2, z_)[::-1].copy()
    cr4 = np.stack([O.synth_adi(6, N_, seed=190 + i)[0] for i in range(z_)]).astype(np.float32)
    g = {"cube": c4, "angles": a4, "scale_list": sc, "cube_ref": cr4}
    kw = dict(scale_list=sc, adimsdi="double", verbose=False, nproc=1, full_output=True)
    for tag, extra in (("rsdi", dict(ncomp=(2, 3), cube_ref=cr4)),
                       ("rsdi_tm", dict(ncomp=(1, 2), cube_ref=cr4, scaling="temp-mean", collapse="mean", mask_center_px=3)),
                       ("rsdi_noadi", dict(ncomp=(2, None), cube_ref=cr4)),
                       ("thr", dict(ncomp=(2, 3), source_xy=(24.0, 20.0), delta_rot=0.5, fwhm=4.0, min_frames_pca=3)),
                       ("thr_ref", dict(ncomp=(2, 4), source_xy=(22.0, 9.0), delta_rot=1.0, fwhm=4.0, min_frames_pca=3,
                                        cube_ref=cr4, max_frames_pca=5)),
                       ("thr_aref", dict(ncomp=(None, 3), source_xy=(8.0, 18.0), delta_rot=0.8, fwhm=4.0,
                                         min_frames_pca=3, cube_ref=cr4, ref_strategy="ARSDI"))):
        fo = ref.pca(c4, a4, **kw, **extra)
        for nm, a in zip(("frame", "chan", "chan_der"), fo):
            g["%s_%s" % (tag, nm)] = np.asarray(a, dtype=np.float32)
    save("g26_msdi_double_ref_thr", **g)


# ---- G27: ADI+mSDI double pass with `cube_sig` (the estimate of the signal goes to the second, ADI stage:
# psfsub/pca_fullfr.py:1395,1409,1447), plain, with a reference cube (RSDI) and under a rotation threshold ------------------
if want("g27"):
    z_, n_, N_ = 4, 12, 32
    c4 = np.stack([O.synth_adi(n_, N_, seed=280 + i)[0] for i in range(z_)]).astype(np.float32)
    a4 = np.linspace(0, 85, n_)
    sc = np.linspace(1.0, 1.2, z_)[::-1].copy()
    cr4 = np.stack([O.synth_adi(6, N_, seed=290 + i)[0] for i in range(z_)]).astype(np.float32)
    rng = np.random.default_rng(2700)
    yy, xx = np.mgrid[:N_, :N_]
    sig = np.stack([0.4 * np.exp(-((yy - 16 - 7 * np.sin(t)) ** 2 + (xx - 16 - 7 * np.cos(t)) ** 2) / 4.0)
                    for t in np.deg2rad(a4)]).astype(np.float32)
    g = {"cube": c4, "angles": a4, "scale_list": sc, "cube_ref": cr4, "cube_sig": sig}
    kw = dict(scale_list=sc, adimsdi="double", verbose=False, nproc=1, full_output=True, cube_sig=sig)
    for tag, extra in (("plain", dict(ncomp=(2, 3))),
                       ("rsdi", dict(ncomp=(2, 3), cube_ref=cr4, scaling="temp-mean")),
                       ("thr", dict(ncomp=(1, 3), source_xy=(23.0, 16.0), delta_rot=0.6, fwhm=4.0, min_frames_pca=3))):
        fo = ref.pca(c4, a4, **kw, **extra)
        for nm, a in zip(("frame", "chan", "chan_der"), fo):
            g["%s_%s" % (tag, nm)] = np.asarray(a, dtype=np.float32)
    save("g27_msdi_double_sig", **g)
