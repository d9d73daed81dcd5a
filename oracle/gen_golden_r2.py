"""Fixtures added in round 2 (G20 ...): outputs of the REAL reference (imported read-only through oracle/_shim.py) frozen
as data under tests/golden/; runs only in the build container:

    python oracle/gen_golden_r2.py
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
from vip_hci.psfsub.utils_pca import pca_annulus  # noqa: E402
from vip_hci.metrics.snr_source import indep_ap_centers  # noqa: E402


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# ---- G20: pca_annulus (psfsub/utils_pca.py:617-755): ADI, RDI, scaling, residual cube, 4-D; and the aperture centres of
# the S/N statistic (metrics/snr_source.py:226-318; the aperture sums themselves need photutils, absent here) ----------
n, N = 16, 48
cube, ang = O.synth_adi(n, N, seed=70)
cref = O.synth_adi(11, N, seed=71)[0]
g = {"cube": cube, "angles": ang, "cube_ref": cref}
g["adi"] = pca_annulus(cube, ang, ncomp=3, annulus_width=8, r_guess=14)
g["adi_mean_tm"] = pca_annulus(cube, ang, ncomp=2, annulus_width=6, r_guess=10, scaling="temp-mean", collapse="mean")
g["rdi"] = pca_annulus(cube, ang, ncomp=4, annulus_width=8, r_guess=13.5, cube_ref=cref)
g["cube_res_der"] = pca_annulus(cube, ang, ncomp=3, annulus_width=8, r_guess=14, collapse=None)
g["cube_res"] = pca_annulus(cube, None, ncomp=3, annulus_width=8, r_guess=14, collapse=None)
c4 = np.stack([O.synth_adi(n, N, seed=72 + i)[0] for i in range(3)])
g["cube4"] = c4
g["ifs"] = pca_annulus(c4, ang, ncomp=[2, 3, 2], annulus_width=8, r_guess=14, collapse="median", collapse_ifs="mean")
fr = np.zeros((N, N))
for i, (xy, fw, ex) in enumerate((((33.2, 29.7), 4.0, False), ((10.0, 12.5), 5.3, True), ((24.0, 40.0), 3.0, False))):
    yy, xx = indep_ap_centers(fr, xy, fw, exclude_negative_lobes=ex)
    g["apc_in_%d" % i] = np.array([xy[0], xy[1], fw, float(ex)])
    g["apc_yy_%d" % i], g["apc_xx_%d" % i] = yy, xx
save("g20_pca_annulus", **g)

# ---- G21: cube_derotate at 512 px (padded length 2048, the C2 shear plan) against the REAL reference in the rot90
# quadrants the C2 pin (angles 0..90 -> q in {0, 3, 4}) never reaches: q = 1, 2 and the half-to-even cases 135 / 225.
# Inputs are reproducible from the seed (not stored); outputs float32.
rng = np.random.default_rng(2100)
fr = (rng.standard_normal((5, 512, 512)) * 3).astype(np.float32)
angs = np.array([-100.0, -135.0, -170.0, -200.1, -225.0])          # cube_derotate rotates by -angle: theta = 100 ... 225
out = ref.cube_derotate(fr, angs, imlib="vip-fft")
save("g21_rotate_512", angles=angs, out=np.asarray(out, dtype=np.float32),
     checksum=np.array([float(np.abs(fr).sum())]))

# ---- G22: the same at 1024 px (padded length 4096, the two-waves-per-line plan of BASELINE configs[4]), one frame per
# rot90 quadrant; a band of 128 rows and 8 columns of every output are kept.
rng = np.random.default_rng(2200)
fr = (rng.standard_normal((4, 1024, 1024)) * 3).astype(np.float32)
angs = np.array([-20.0, -100.0, -200.1, -290.0])
out = np.asarray(ref.cube_derotate(fr, angs, imlib="vip-fft"))
save("g22_rotate_1024", angles=angs, band=out[:, 448:576, :].astype(np.float32), cols=out[:, :, 500:508].astype(np.float32))

# ---- G23: ADI+mSDI single-pass leftovers (psfsub/pca_fullfr.py:1038-1243): grid of PCs (tuple / list ncomp ->
# pca_grid with scale_list, utils_pca.py:201-227), with and without S/N scoring at source_xy, and a reference cube ----
if "g23" in sys.argv or len(sys.argv) == 1:
    z_, n_, N_ = 4, 8, 32
    c4 = np.stack([O.synth_adi(n_, N_, seed=80 + i)[0] for i in range(z_)]).astype(np.float32)
    a4 = np.linspace(0, 70, n_)
    sc = np.linspace(1.0, 1.2, z_)[::-1].copy()
    cr4 = np.stack([O.synth_adi(5, N_, seed=90 + i)[0] for i in range(z_)]).astype(np.float32)
    g = {"cube": c4, "angles": a4, "scale_list": sc, "cube_ref": cr4}
    g["grid_frames"] = np.asarray(ref.pca(c4, a4, scale_list=sc, adimsdi="single", ncomp=(1, 4), verbose=False, nproc=1))
    fo = ref.pca(c4, a4, scale_list=sc, adimsdi="single", ncomp=[2, 5], full_output=True, verbose=False, nproc=1)
    g["list_frames"], g["list_pcs"] = np.asarray(fo[0]), np.asarray(fo[1])
    g["grid_range_median"] = np.asarray(ref.pca(c4, a4, scale_list=sc, adimsdi="single", ncomp=(1, 5, 2), verbose=False,
                                                nproc=1, ifs_collapse_range=(1, 4), collapse="mean",
                                                scaling="temp-mean", mask_center_px=3))
    fo = ref.pca(c4, a4, scale_list=sc, adimsdi="single", ncomp=3, cube_ref=cr4, full_output=True, verbose=False,
                 nproc=1)
    for nm, a in zip(("frame", "allfr", "desc", "adi"), fo):
        g["ref_%s" % nm] = np.asarray(a)
    save("g23_msdi_single_more", **g)

# ---- G24: left_eigv=True (psfsub/pca_fullfr.py:428-437,1720-1724: PCs in the temporal domain, ADI only): the residuals
# equal those of the standard projection; full_output returns pcs = U^T (k x n) -------------------------------------
if "g24" in sys.argv or len(sys.argv) == 1:
    cube, ang = O.synth_adi(14, 40, seed=95)
    g = {"cube": cube, "angles": ang}
    fo = ref.pca(cube, ang, ncomp=3, left_eigv=True, full_output=True, verbose=False, nproc=1)
    for nm, a in zip(("frame", "pcs", "recon", "res", "resd"), fo):
        g["left_" + nm] = np.asarray(a)
    g["left_frame_only"] = np.asarray(ref.pca(cube, ang, ncomp=3, left_eigv=True, verbose=False, nproc=1))
    g["std_frame"] = np.asarray(ref.pca(cube, ang, ncomp=3, verbose=False, nproc=1))
    from vip_hci.psfsub.svd import svd_wrapper as _svdw
    M = cube.reshape(14, -1).astype(np.float64)
    g["svd_left_lapack"] = _svdw(M, "lapack", 4, False, left_eigv=True)
    g["svd_left_arpack"] = _svdw(M, "arpack", 4, False, left_eigv=True)
    save("g24_left_eigv", **g)
