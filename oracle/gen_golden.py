"""Freeze outputs of the REAL reference (vip_hci, imported read-only through
oracle/_shim.py) into small fixtures under tests/golden/.  Runs only in the build container:

    python oracle/gen_golden.py

Fixtures are data only (seeded inputs + the reference's outputs); no reference source text
is stored.  tests/test_oracle_golden.py checks oracle/ref_cpu.py against them on any box,
and the ``-m gpu`` parity tests check the HIP path against them.
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
os.makedirs(OUT, exist_ok=True)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


rng = np.random.default_rng(2024)

# G1 svd_wrapper --------------------------------------------------------------------------------
g = {}
for tag, shape in (("a", (20, 100)), ("b", (50, 4096))):
    M = rng.standard_normal(shape).astype(np.float32)
    M += (np.outer(rng.standard_normal(shape[0]), rng.standard_normal(shape[1])) * 4).astype(np.float32)
    M += (np.outer(rng.standard_normal(shape[0]), rng.standard_normal(shape[1])) * 2).astype(np.float32)
    g["M_" + tag] = M
    for mode in ("lapack", "eigen"):
        U, S, V = ref.svd_wrapper(M, mode, 6, False, full_output=True)
        g["V_%s_%s" % (mode, tag)] = V
        g["S_%s_%s" % (mode, tag)] = S
save("g1_svd", **g)

# G2 project_subtract ---------------------------------------------------------------------------
cube, ang = O.synth_adi(12, 32, seed=3)
g = {"cube": cube, "angles": ang}
for sc in (None, "temp-mean", "temp-standard", "spat-mean", "spat-standard"):
    for mk in (None, 4):
        r = ref._project_subtract(cube, None, 3, sc, mk, "lapack", False, False)
        g["res_%s_%s" % (sc, mk)] = r
save("g2_project_subtract", **g)

# G3 frame_rotate ---------------------------------------------------------------------------------
ANGLES = np.array([-370, -30, 0, 12.5, 44.9, 45, 45.1, 90, 135, 135.3, 180, 271, 315, 359.9, 360])
g = {"angles": ANGLES}
for N in (32, 33, 64):
    fr = rng.standard_normal((N, N)).astype(np.float32)
    g["frame_%d" % N] = fr
    g["rot_%d" % N] = np.stack([ref.frame_rotate(fr, th, imlib="vip-fft") for th in ANGLES])
frn = rng.standard_normal((40, 40)).astype(np.float32)
frn[5:8, 9] = np.nan
g["frame_nan"] = frn
g["rot_nan"] = ref.frame_rotate(frn, 33.0)
fr0 = rng.standard_normal((40, 40)).astype(np.float32)
fr0[18:23, 18:23] = 0
g["frame_zero"] = fr0
g["rot_zero"] = ref.frame_rotate(fr0, 33.0, mask_val=0, interp_zeros=True, ker=1)
cube, ang = O.synth_adi(6, 33, seed=5)
g["cube_33"] = cube
g["cube_33_angles"] = ang
g["derot_33"] = ref.cube_derotate(cube, ang, nproc=1)
cube, ang = O.synth_adi(5, 128, seed=9)       # power-of-two padded length (L=512): FFT fast path
g["cube_128"] = cube
g["cube_128_angles"] = np.array([3.0, 47.5, 95.0, 200.1, 333.3])
g["derot_128"] = ref.cube_derotate(cube, g["cube_128_angles"], nproc=1)
save("g3_rotate", **g)

# G4 collapse -----------------------------------------------------------------------------------
g = {}
for n in (7, 8):
    cb = rng.standard_normal((n, 9, 11)).astype(np.float32)
    cbn = cb.copy()
    cbn[1, 2, 3] = np.nan
    cbn[:, 4, 4] = np.nan
    cbn[0:5, 0, 0] = np.nan
    g["cube_%d" % n] = cb
    g["cube_nan_%d" % n] = cbn
    for mode in ("median", "mean", "sum", "max", "absmean"):
        g["%s_%d" % (mode, n)] = ref.cube_collapse(cb, mode)
        g["%s_nan_%d" % (mode, n)] = ref.cube_collapse(cbn, mode)
    w = rng.random(n)
    g["w_%d" % n] = w
    g["wmean_%d" % n] = ref.cube_collapse(cb.copy(), "wmean", w=w)
    g["trimmean_%d" % n] = ref.cube_collapse(cb, "trimmean", n=3)
save("g4_collapse", **g)

# G5 index sets -----------------------------------------------------------------------------------
g = {}
angles = np.array([130, 120, 90, 60, 30, 10, 0.])
g["fi_angles"] = angles
for fr_i in range(7):
    g["fi_%d" % fr_i] = ref._find_indices_adi(angles, fr_i, 42)
ang_long = np.linspace(0, 80, 60)
g["fi_long_angles"] = ang_long
for fr_i in (0, 7, 30, 59):
    for mf in (10, 25):
        g["fit_%d_%d" % (fr_i, mf)] = ref._find_indices_adi(ang_long, fr_i, 3.0, truncate=True, max_frames=mf)
for (N, inner, w, ns, th0) in ((64, 8, 8, 1, 0), (65, 7, 8, 3, 30), (512, 223, 32, 1, 0)):
    e = ref.get_annulus_segments(np.zeros((N, N)), inner, w, ns, th0)
    for i in range(ns):
        g["seg_%d_%d_%d_%d_%d_s%d" % (N, inner, w, ns, th0, i)] = (e[i][0].astype(np.int64) * N + e[i][1]).astype(np.int32)
g["define_annuli"] = np.array([ref._define_annuli(ang_long, ann, 4, 4, 2, 6, 0.5, 1, False, True) for ann in range(4)])
for a_i, a in enumerate(([10., 20, 30], [-10., 5, 20], [350., 355, 0, 5], [170., 190, 10])):
    g["pa_in_%d" % a_i] = np.array(a)
    g["pa_out_%d" % a_i] = ref.check_pa_vector(np.array(a))
m3 = rng.standard_normal((2, 21, 21)).astype(np.float32)
g["mask_in"] = m3
g["mask_out_5"] = ref.mask_circle(m3, 5)
save("g5_indices", **g)

# G6 end-to-end ----------------------------------------------------------------------------------
cube, ang = O.synth_adi(50, 128, seed=0)          # C1 scale
fr = ref.pca(cube, ang, ncomp=5, full_output=True, verbose=False)
c0 = 64 - 8
save("g6_pca_c1", seed=0, frame=fr[0], res_crop=fr[3][:, c0:c0 + 16, c0:c0 + 16],
     res_sum=np.sum(fr[3].astype(np.float64), axis=0), resder_crop=fr[4][:, c0:c0 + 16, c0:c0 + 16],
     resder_sum=np.sum(fr[4].astype(np.float64), axis=0))

cube, ang = O.synth_adi(16, 40, seed=6)
g = {"cube": cube, "angles": ang}
for tag, kw in (("k3", dict(ncomp=3)), ("eigen", dict(ncomp=3, svd_mode="eigen")),
                ("tmean", dict(ncomp=2, scaling="temp-mean")), ("tstd", dict(ncomp=2, scaling="temp-standard")),
                ("smean", dict(ncomp=2, scaling="spat-mean")), ("sstd", dict(ncomp=2, scaling="spat-standard")),
                ("mask", dict(ncomp=3, mask_center_px=5)), ("mean", dict(ncomp=4, collapse="mean"))):
    fo = ref.pca(cube, ang, full_output=True, verbose=False, **kw)
    for nm, a in zip(("frame", "pcs", "recon", "res", "resder"), fo):
        g["%s_%s" % (tag, nm)] = a
cref, _ = O.synth_adi(10, 40, seed=7)
g["cube_ref"] = cref
g["rdi_frame"] = ref.pca(cube, ang, cube_ref=cref, ncomp=3, verbose=False)
save("g6_pca_small", **g)

c4 = np.stack([O.synth_adi(10, 32, seed=10 + i)[0] for i in range(3)])
a4 = np.linspace(0, 70, 10)
fo = ref.pca(c4, a4, ncomp=2, full_output=True, verbose=False)
save("g6_pca_4d", cube=c4, angles=a4, frame=fo[0], res=fo[3], resder=fo[4], ifs=fo[5])

cube, ang = O.synth_adi(30, 64, seed=8)
g = {"cube": cube, "angles": ang}
for tag, kw in (("a", dict(asize=8, ncomp=3, fwhm=4, delta_rot=(0.1, 1))),
                ("b", dict(asize=8, ncomp=2, fwhm=4, delta_rot=0.5, radius_int=4, max_frames_lib=12)),
                ("c", dict(asize=10, ncomp=(1, 2, 3), fwhm=4, delta_rot=(0.1, 1), n_segments=2))):
    cr, cdr, fr_ = ref.pca_annular(cube, ang, full_output=True, verbose=False, nproc=1, **kw)
    g[tag + "_cube_out"] = cr
    g[tag + "_frame"] = fr_
save("g6_pca_annular", **g)

# ---- G7: pca_grid (tuple / list ncomp) and PA-threshold frame rejection (source_xy) ----------------------------
cube, ang = O.synth_adi(24, 48, seed=11)
g = {"cube": cube, "angles": ang}
out = ref.pca(cube, ang, ncomp=(1, 5), full_output=True, verbose=False)
g["grid_a_frames"], g["grid_a_pcs"] = out[0], np.asarray(out[1])
out = ref.pca(cube, ang, ncomp=(2, 9, 3), scaling="temp-mean", mask_center_px=4, verbose=False)
g["grid_b_frames"] = out
out = ref.pca(cube, ang, ncomp=[1, 4, 6], collapse="mean", verbose=False)
g["grid_c_frames"] = out
cref, _ = O.synth_adi(10, 48, seed=12)
g["cube_ref"] = cref
g["grid_d_frames"] = ref.pca(cube, ang, cube_ref=cref, ncomp=(1, 4), verbose=False)
for tag, kw in (("rej_a", dict(ncomp=3, source_xy=(34, 24), fwhm=4, delta_rot=1, min_frames_pca=4)),
                ("rej_b", dict(ncomp=2, source_xy=(30, 30), fwhm=4, delta_rot=0.5, min_frames_pca=3, max_frames_pca=8,
                               scaling="temp-standard")),
                ("rej_c", dict(ncomp=2, source_xy=(10, 24), fwhm=5, delta_rot=1, min_frames_pca=2, mask_center_px=3))):
    fo = ref.pca(cube, ang, full_output=True, verbose=False, **kw)
    for nm, a in zip(("frame", "recon", "res", "resder"), fo):
        g["%s_%s" % (tag, nm)] = a
save("g7_grid_rejection", **g)

# ---- G8: median_sub (full-frame) and STIM maps --------------------------------------------------------------------
cube, ang = O.synth_adi(16, 40, seed=13)
g = {"cube": cube, "angles": ang}
for tag, kw in (("a", dict()), ("b", dict(radius_int=4, collapse="mean")),
                ("c", dict(cube_ref=O.synth_adi(9, 40, seed=14)[0], collapse_ref="median")),
                ("d", dict(cube_ref=O.synth_adi(9, 40, seed=14)[0], collapse_ref="mean"))):
    co, cd, fr_ = ref.median_sub(cube, ang, full_output=True, verbose=False, nproc=1, **kw)
    g["ms_%s_out" % tag], g["ms_%s_der" % tag], g["ms_%s_frame" % tag] = co, cd, fr_
g["cube_ref"] = O.synth_adi(9, 40, seed=14)[0]
res = ref.pca(cube, ang, ncomp=3, full_output=True, verbose=False)
g["res"], g["resder"] = res[3], res[4]
g["stim"] = ref.stim_map(res[4])
g["stim_inv"] = ref.inverse_stim_map(res[3], ang)
g["stim_norm"] = ref.normalized_stim_map(res[3], ang)
g["stim_norm_mask"] = ref.normalized_stim_map(res[3], ang, mask=5)
save("g8_medsub_stim", **g)

# ---- G9: pca_annular on a 4-D cube without scale_list (per-channel loop) -----------------------------------------
c4 = np.stack([O.synth_adi(12, 40, seed=40 + i)[0] for i in range(3)])
a4 = np.linspace(0, 80, 12)
co, cd, fr_ = ref.pca_annular(c4, a4, asize=8, ncomp=2, fwhm=4, delta_rot=(0.1, 1), full_output=True, verbose=False,
                              nproc=1)
save("g9_annular_4d", cube=c4, angles=a4, cube_out=co, cube_der=cd, frame=fr_)

# ---- G10: ADI+mSDI (4-D cube + scale_list): rescaling of the channels, double- and single-pass PCA ----------------
z_, n_, N_ = 5, 8, 32
c4 = np.stack([O.synth_adi(n_, N_, seed=50 + i)[0] for i in range(z_)]).astype(np.float32)
a4 = np.linspace(0, 60, n_)
sc = np.linspace(1.0, 1.25, z_)[::-1].copy()
g = {"cube": c4, "angles": a4, "scale_list": sc}
from vip_hci.preproc import cube_rescaling_wavelengths as _scw          # noqa: E402
r = _scw(c4[:, 0].astype(np.float64), sc, imlib="vip-fft")
g["scw_cube"], g["scw_frame"] = r[0], r[1]
ri = _scw(r[0], sc, full_output=True, inverse=True, y_in=N_, x_in=N_, imlib="vip-fft", collapse="mean")
g["scw_inv_cube"], g["scw_inv_frame"] = ri[0], ri[1]
for tag, kw in (("d22", dict(ncomp=(2, 2))), ("dN2", dict(ncomp=(None, 2))), ("d2N", dict(ncomp=(2, None))),
                ("dmask", dict(ncomp=(2, 2), mask_center_px=3, scaling="temp-mean", collapse_ifs="median")),
                ("drange", dict(ncomp=(2, 2), ifs_collapse_range=(1, 4)))):
    fo = ref.pca(c4, a4, scale_list=sc, adimsdi="double", full_output=True, verbose=False, nproc=1, **kw)
    for nm, a in zip(("frame", "rcc", "rcc_der"), fo):
        g["%s_%s" % (tag, nm)] = np.asarray(a)
for tag, kw in (("s3", dict(ncomp=3)), ("s3nocrop", dict(ncomp=3, crop_ifs=False)),
                ("s2mask", dict(ncomp=2, mask_center_px=3, scaling="temp-standard"))):
    fo = ref.pca(c4, a4, scale_list=sc, adimsdi="single", full_output=True, verbose=False, nproc=1, **kw)
    for nm, a in zip(("frame", "allfr", "desc", "adi"), fo):
        g["%s_%s" % (tag, nm)] = np.asarray(a)
save("g10_msdi", **g)
