"""Fixtures added after the first set (G11 ...): same rules as oracle/gen_golden.py -- outputs of the REAL reference
(imported read-only through oracle/_shim.py) frozen as data under tests/golden/; runs only in the build container:

    python oracle/gen_golden_more.py
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


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# ---- G11: cube_sig (estimated-signal cube: PCs learnt from / projection of cube - cube_sig, model subtracted from
# the cube; reference pca_fullfr.py:1652-1662,1717-1731) in the whole-matrix, RDI and source_xy branches ------------
n, N = 14, 36
cube, ang = O.synth_adi(n, N, seed=60)
ang = np.linspace(0, 70, n)
yy, xx = np.mgrid[:N, :N]
sig = np.zeros_like(cube)
for i, a in enumerate(np.deg2rad(ang)):          # a faint companion moving with the parallactic angle
    cy, cx = N // 2 + 9 * np.sin(a), N // 2 + 9 * np.cos(a)
    sig[i] = 0.8 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.7 ** 2))
cube = (cube + sig).astype(np.float32)
sig = (0.9 * sig).astype(np.float32)             # an imperfect estimate of it
cref = O.synth_adi(9, N, seed=61)[0]
g = {"cube": cube, "angles": ang, "cube_sig": sig, "cube_ref": cref}
for tag, kw in (("plain", dict(ncomp=3)), ("scaled", dict(ncomp=2, scaling="temp-mean", mask_center_px=3)),
                ("rdi", dict(ncomp=3, cube_ref=cref))):
    fo = ref.pca(cube, ang, cube_sig=sig, full_output=True, verbose=False, nproc=1, **kw)
    for nm, a in zip(("frame", "pcs", "recon", "res", "resder"), fo):
        g["%s_%s" % (tag, nm)] = np.asarray(a)
fo = ref.pca(cube, ang, ncomp=2, cube_sig=sig, source_xy=(N // 2 + 9, N // 2), fwhm=4, delta_rot=1, min_frames_pca=3,
             full_output=True, verbose=False, nproc=1)
for nm, a in zip(("frame", "recon", "res", "resder"), fo):
    g["sxy_%s" % nm] = np.asarray(a)
save("g11_cube_sig", **g)

# ---- G12: pca_annular with a LIST of ncomp (several truncations of one decomposition; pca_local.py:665-668,892-902) --
cube, _ = O.synth_adi(16, 40, seed=70)
ang = np.linspace(0, 85, 16)
g = {"cube": cube, "angles": ang}
for tag, kw in (("a", dict(ncomp=[1, 3, 6], asize=8, fwhm=4, delta_rot=(0.1, 1))),
                ("b", dict(ncomp=[2, 4], asize=5, fwhm=4, delta_rot=0.5, n_segments=2, radius_int=5,
                           scaling="temp-standard", collapse="mean"))):
    co, cd, fr_ = ref.pca_annular(cube, ang, full_output=True, verbose=False, nproc=1, **kw)
    g[tag + "_out"], g[tag + "_der"], g[tag + "_frames"] = co.astype(np.float32), cd.astype(np.float32), np.stack(fr_)
    g[tag + "_dtypes"] = np.array([str(co.dtype), str(cd.dtype), str(fr_[0].dtype)])
save("g12_annular_list", **g)

# ---- G13: pca_annular with a reference cube (RDI), an estimated-signal cube, and both (pca_local.py:716-724,862-891) ---
cube, _ = O.synth_adi(14, 40, seed=80)
ang = np.linspace(0, 80, 14)
cref = O.synth_adi(8, 40, seed=81)[0]
yy, xx = np.mgrid[:40, :40]
sig = np.stack([0.7 * np.exp(-((yy - 20 - 9 * np.sin(a)) ** 2 + (xx - 20 - 9 * np.cos(a)) ** 2) / (2 * 1.7 ** 2))
                for a in np.deg2rad(ang)]).astype(np.float32)
cube = (cube + sig).astype(np.float32)
sig = (0.9 * sig).astype(np.float32)
g = {"cube": cube, "angles": ang, "cube_ref": cref, "cube_sig": sig}
for tag, kw in (("ref", dict(cube_ref=cref, ncomp=3, asize=8, fwhm=4, delta_rot=(0.1, 1))),
                ("sig", dict(cube_sig=sig, ncomp=2, asize=8, fwhm=4, delta_rot=0.5, scaling="temp-mean")),
                ("both", dict(cube_ref=cref, cube_sig=sig, ncomp=[2, 5], asize=10, fwhm=4, delta_rot=1, n_segments=2))):
    co, cd, fr_ = ref.pca_annular(cube, ang, full_output=True, verbose=False, nproc=1, **kw)
    g[tag + "_out"], g[tag + "_der"] = np.asarray(co, dtype=np.float32), np.asarray(cd, dtype=np.float32)
    g[tag + "_frame"] = np.stack(fr_) if isinstance(fr_, list) else fr_
save("g13_annular_ref_sig", **g)

# ---- G14: median_sub(mode='annular'): ADI with nframes (default 4 and 2), radius_int, and RDI (medsub.py:316-371,602-676) --
cube, _ = O.synth_adi(18, 44, seed=90)
ang = np.linspace(0, 75, 18)
cref = O.synth_adi(7, 44, seed=91)[0]
g = {"cube": cube, "angles": ang, "cube_ref": cref}
for tag, kw in (("a", dict(asize=4, fwhm=4, delta_rot=1)), ("b", dict(asize=6, fwhm=3, delta_rot=0.5, nframes=2, collapse="mean")),
                ("c", dict(asize=5, fwhm=4, radius_int=4, nframes=6)), ("d", dict(asize=4, cube_ref=cref, collapse_ref="mean"))):
    co, cd, fr_ = ref.median_sub(cube, ang, mode="annular", full_output=True, verbose=False, nproc=1, **kw)
    g["ms_%s_out" % tag], g["ms_%s_der" % tag], g["ms_%s_frame" % tag] = co, cd, fr_
save("g14_medsub_annular", **g)

# ---- G18: ADI+mSDI at a larger / odd frame size (zoom operators for odd and even sizes, more channels) ---------------
for N_ in (64, 65):
    z_, n_ = 7, 12
    c4 = np.stack([O.synth_adi(n_, N_, seed=100 + i)[0] for i in range(z_)]).astype(np.float32)
    a4 = np.linspace(0, 70, n_)
    sc = np.linspace(1.0, 1.4, z_)[::-1].copy()
    g = {"cube": c4, "angles": a4, "scale_list": sc}
    fo = ref.pca(c4, a4, scale_list=sc, adimsdi="double", ncomp=(2, 3), full_output=True, verbose=False, nproc=1)
    for nm, a in zip(("frame", "rcc", "rcc_der"), fo):
        g["d_%s" % nm] = np.asarray(a)
    fo = ref.pca(c4, a4, scale_list=sc, adimsdi="single", ncomp=4, full_output=True, verbose=False, nproc=1)
    g["s_frame"] = np.asarray(fo[0])
    g["s_adi"] = np.asarray(fo[3])
    save("g18_msdi_%d" % N_, **g)

# ---- G19: rarely used switches: pca_annular(n_segments='auto', theta_init, radius_int, max_frames_lib), pca(collapse='wmean') --
cube, _ = O.synth_adi(20, 48, seed=110)
ang = np.linspace(0, 85, 20)
w = np.linspace(0.5, 1.5, 20)
g = {"cube": cube, "angles": ang, "weights": w}
co, cd, fr_ = ref.pca_annular(cube, ang, ncomp=2, asize=6, fwhm=4, delta_rot=(0.1, 0.8), n_segments="auto", theta_init=30,
                              radius_int=6, max_frames_lib=9, full_output=True, verbose=False, nproc=1)
g["ann_out"], g["ann_der"], g["ann_frame"] = co, cd, fr_
g["pca_wmean"] = ref.pca(cube, ang, ncomp=3, collapse="wmean", weights=w, verbose=False, nproc=1)
g["ann_wmean"] = ref.pca_annular(cube, ang, ncomp=2, asize=8, fwhm=4, collapse="wmean", weights=w, verbose=False, nproc=1)
save("g19_switches", **g)
