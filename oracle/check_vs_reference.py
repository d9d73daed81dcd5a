"""Pin the CPU oracle (oracle/ref_cpu.py) against the real reference, imported read-only
from /root/reference via oracle/_shim.py.  Runs ONLY in the build container.

    python oracle/check_vs_reference.py

Also replays the reference's own network-free known-answer tests for this path
(tests/pre_3_10/test_pca_svd.py, test_preproc_rotation.py) against the oracle.
"""
import sys
import os
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _shim, ref_cpu as O  # noqa: E402

warnings.simplefilter("ignore")
ref = _shim.load()
FAILS = []


def check(name, got, exp, atol=0.0, rtol=0.0, exact=False):
    got = np.asarray(got)
    exp = np.asarray(exp)
    if got.shape != exp.shape:
        FAILS.append(name)
        print("FAIL %-48s shape %s vs %s" % (name, got.shape, exp.shape))
        return
    if exact:
        ok = np.array_equal(got, exp, equal_nan=True)
        err = 0 if ok else 1
    else:
        d = np.abs(got.astype(float) - exp.astype(float))
        d = np.where(np.isnan(got) & np.isnan(exp), 0, d)
        err = float(np.nanmax(d)) if d.size else 0.0
        bound = atol + rtol * np.abs(np.nan_to_num(exp.astype(float)))
        ok = bool(np.all(d <= bound)) and not np.any(np.isnan(d))
    print("%s %-48s err=%.3g" % ("ok  " if ok else "FAIL", name, err))
    if not ok:
        FAILS.append(name)


def sign_align(V, Vref):
    s = np.sign(np.sum(V * Vref, axis=1))
    s[s == 0] = 1
    return V * s[:, None]


rng = np.random.default_rng(1)

# --- geometry / host helpers -----------------------------------------------------------
for N in (32, 33, 64, 65, 80, 81, 100, 101, 128, 129, 256, 511, 512):
    fr = np.zeros((N, N))
    n1 = int(N * 1.5)
    if n1 % 2 != N % 2:
        n1 += 1
    pad, idx = ref.frame_pad(np.zeros((n1, n1)), fac=4 / 1.5, fillwith=0, full_output=True)
    L, Le, off = O.rot_geometry(N)
    cy = n1 // 2
    y0p = int(cy - N // 2)
    check("rot_geometry L N=%d" % N, [L, idx[0] + y0p], [pad.shape[0], off], exact=True)

for a in ([10., 20, 30], [-10., 5, 20], [350., 355, 0, 5], [170., 190, 10], [-5., -170, 100]):
    a = np.array(a)
    check("check_pa_vector %s" % a[:2], O.check_pa_vector(a), ref.check_pa_vector(a), exact=True)

angles = np.array([130, 120, 90, 60, 30, 10, 0.])
for fr_i in range(7):
    for thr in (42, 15, 200):
        check("find_indices_adi f=%d thr=%d" % (fr_i, thr), O.find_indices_adi(angles, fr_i, thr),
              ref._find_indices_adi(angles, fr_i, thr), exact=True)
ang_long = np.linspace(0, 80, 60)
for fr_i in (0, 7, 30, 59):
    for mf in (10, 25, 200):
        check("find_indices_adi trunc f=%d mf=%d" % (fr_i, mf),
              O.find_indices_adi(ang_long, fr_i, 3.0, truncate=True, max_frames=mf),
              ref._find_indices_adi(ang_long, fr_i, 3.0, truncate=True, max_frames=mf), exact=True)

for ann in range(4):
    got = O.define_annuli(ang_long, ann, 4, 4, 2, 6, 0.5, strict=True)
    exp = ref._define_annuli(ang_long, ann, 4, 4, 2, 6, 0.5, 1, False, True)
    check("define_annuli ann=%d" % ann, got, exp, rtol=1e-15)

for (N, inner, w, ns, th0) in ((64, 8, 8, 1, 0), (65, 0, 8, 3, 0), (65, 7, 8, 3, 30), (101, 12, 4, 4, 200),
                               (128, 31, 32, 1, 0), (512, 223, 32, 1, 0)):
    g = O.get_annulus_segments((N, N), inner, w, ns, th0)
    e = ref.get_annulus_segments(np.zeros((N, N)), inner, w, ns, th0)
    for i in range(ns):
        check("annulus_segments N=%d in=%d seg=%d y" % (N, inner, i), g[i][0], e[i][0], exact=True)
        check("annulus_segments N=%d in=%d seg=%d x" % (N, inner, i), g[i][1], e[i][1], exact=True)

for N, r in ((21, 5), (32, 4), (33, 6.5)):
    a = rng.standard_normal((3, N, N)).astype(np.float32)
    check("mask_circle N=%d r=%s" % (N, r), O.mask_circle(a, r), ref.mask_circle(a, r), exact=True)

# --- scaling -----------------------------------------------------------------------------
m32 = (rng.standard_normal((12, 200)) * 3 + 5).astype(np.float32)
m32[:, 7] = 2.5      # constant column
for sc in ("temp-mean", "temp-standard", "spat-mean", "spat-standard"):
    check("matrix_scaling %s f32" % sc, O.matrix_scaling(m32, sc), ref.matrix_scaling(m32, sc), exact=True)
    check("matrix_scaling %s f64" % sc, O.matrix_scaling(m32.astype(float), sc),
          ref.matrix_scaling(m32.astype(float), sc), exact=True)

# --- svd_wrapper ---------------------------------------------------------------------------
mat = np.random.RandomState(42).randn(20, 100)
U, S, V = O.svd_wrapper(mat, "lapack", 20, full_output=True)
check("ref test_svd_recons (oracle)", np.abs(U @ np.diag(S) @ V), np.abs(mat), atol=1e-2)
for shape in ((20, 100), (50, 4096)):
    M = rng.standard_normal(shape).astype(np.float32)
    M += np.outer(rng.standard_normal(shape[0]), rng.standard_normal(shape[1])).astype(np.float32) * 5
    for mode in ("lapack", "eigen"):
        Vo = O.svd_wrapper(M, mode, 5)
        Vr = ref.svd_wrapper(M, mode, 5, False)
        check("svd_wrapper %s %s" % (mode, shape), sign_align(Vo, Vr), Vr, atol=2e-5)
        Uo, So, Vo2 = O.svd_wrapper(M, mode, 5, full_output=True)
        Ur, Sr, Vr2 = ref.svd_wrapper(M, mode, 5, False, full_output=True)
        check("svd_wrapper %s %s S" % (mode, shape), So, Sr, rtol=1e-5)
        check("svd_wrapper %s %s U-shape" % (mode, shape), np.array(Uo.shape), np.array(Ur.shape), exact=True)

# --- project / subtract ------------------------------------------------------------------
cube, ang = O.synth_adi(12, 32, seed=3)
for sc in (None, "temp-mean", "temp-standard", "spat-mean", "spat-standard"):
    for mk in (None, 4):
        r_o = O.project_subtract(cube, 3, sc, mk, "lapack")
        r_r = ref._project_subtract(cube, None, 3, sc, mk, "lapack", False, False)
        check("project_subtract sc=%s mask=%s" % (sc, mk), r_o, r_r, atol=5e-5)
r_o = O.project_subtract(cube, 0.9, None, None, "lapack")
r_r = ref._project_subtract(cube, None, 0.9, None, None, "lapack", False, False)
check("project_subtract cevr=0.9", r_o, r_r, atol=5e-5)
cube_ref, _ = O.synth_adi(9, 32, seed=4)
r_o = O.project_subtract(cube, 3, None, None, "lapack", cube_ref=cube_ref)
r_r = ref._project_subtract(cube, cube_ref, 3, None, None, "lapack", False, False)
check("project_subtract RDI", r_o, r_r, atol=5e-5)

# --- rotation -------------------------------------------------------------------------------
ANGLES = (-370, -30, 0, 12.5, 44.9, 45, 45.1, 90, 135, 135.3, 180, 271, 315, 359.9, 360, 725.5)
for N in (32, 33, 64, 65):
    fr = rng.standard_normal((N, N))
    for th in ANGLES:
        check("frame_rotate N=%d th=%s" % (N, th), O.frame_rotate_fft(fr, th),
              ref.frame_rotate(fr, th, imlib="vip-fft"), atol=1e-12)
fr = rng.standard_normal((40, 40))
fr[5:8, 9] = np.nan
check("frame_rotate NaN mask", O.frame_rotate_fft(fr, 33.0), ref.frame_rotate(fr, 33.0), atol=1e-12)
fr0 = rng.standard_normal((40, 40))
fr0[18:23, 18:23] = 0
check("frame_rotate mask_val=0", O.frame_rotate_fft(fr0, 33.0, mask_val=0),
      ref.frame_rotate(fr0, 33.0, mask_val=0, interp_zeros=True, ker=1), atol=1e-12)
# delta-function known answers
for N in (64, 65):
    d = np.zeros((N, N))
    c = N // 2
    d[c, c + 10] = 1
    for th, (dy, dx, val) in {90: (-10, 0, 1.0), 180: (0, -10, 1.0), 270: (10, 0, 1.0),
                              30: (-5, 9, 0.8207), -30: (5, 9, 0.8207), 45: (-7, 7, 0.9753)}.items():
        o = O.frame_rotate_fft(d, th)
        iy, ix = np.unravel_index(np.argmax(o), o.shape)
        check("delta N=%d th=%d" % (N, th), [iy - c, ix - c, round(o[iy, ix], 3)], [dy, dx, round(val, 3)],
              atol=6e-4)

# reference test_cube_derotate: ones cube, 24 successive derotations return to the input
for N in (80, 81):
    arr = np.ones((4, N, N))
    angs = np.array([120, 90, 60, 45.])
    cur = arr.copy()
    for _ in range(24):
        cur = O.cube_derotate(cur, angs)
    c0 = N // 2 - 25
    check("ref test_cube_derotate N=%d (oracle)" % N, cur[:, c0:c0 + 50, c0:c0 + 50],
          arr[:, c0:c0 + 50, c0:c0 + 50], atol=1e-1, rtol=1e-1)

cube, ang = O.synth_adi(6, 33, seed=5)
check("cube_derotate f32", O.cube_derotate(cube, ang), ref.cube_derotate(cube, ang, nproc=1), atol=1e-6)

# --- collapse ----------------------------------------------------------------------------------
for n in (7, 8):
    cb = rng.standard_normal((n, 9, 11)).astype(np.float32)
    cbn = cb.copy()
    cbn[1, 2, 3] = np.nan
    cbn[:, 4, 4] = np.nan
    cbn[0:5, 0, 0] = np.nan
    for mode in ("median", "mean", "sum", "max", "absmean"):
        check("collapse %s n=%d" % (mode, n), O.cube_collapse(cb, mode), ref.cube_collapse(cb, mode), exact=True)
        check("collapse %s n=%d nan" % (mode, n), O.cube_collapse(cbn, mode), ref.cube_collapse(cbn, mode), exact=True)
    w = rng.random(n)
    check("collapse wmean n=%d" % n, O.cube_collapse(cb, "wmean", w=w), ref.cube_collapse(cb.copy(), "wmean", w=w),
          rtol=1e-6, atol=1e-7)
    check("collapse trimmean n=%d" % n, O.cube_collapse(cb, "trimmean", n=3),
          ref.cube_collapse(cb, "trimmean", n=3), rtol=1e-6, atol=1e-7)

# --- end to end --------------------------------------------------------------------------------
cube, ang = O.synth_adi(16, 40, seed=6)
for kw in (dict(ncomp=3), dict(ncomp=3, svd_mode="eigen"), dict(ncomp=2, scaling="temp-mean"),
           dict(ncomp=3, mask_center_px=5), dict(ncomp=4, collapse="mean"), dict(ncomp=40)):
    fo = O.pca_fullframe(cube, ang, full_output=True, **kw)
    fr = ref.pca(cube, ang, full_output=True, verbose=False, **kw)
    names = ("frame", "pcs", "recon", "res", "res_der")
    for nm, a, b in zip(names, fo, fr):
        if nm == "pcs":
            a = sign_align(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)).reshape(b.shape)
            # flat part of the spectrum (ncomp=n) is only defined up to the subspace
            if kw.get("ncomp") == 40:
                continue
        check("pca %s %s" % (kw, nm), a, b, atol=1e-4 if nm != "recon" else 5e-4)
    check("pca %s frame(no full_output)" % kw, O.pca_fullframe(cube, ang, **kw),
          ref.pca(cube, ang, verbose=False, **kw), atol=1e-4)

cref, _ = O.synth_adi(10, 40, seed=7)
check("pca RDI", O.pca_fullframe(cube, ang, ncomp=3, cube_ref=cref),
      ref.pca(cube, ang, cube_ref=cref, ncomp=3, verbose=False), atol=1e-4)

c4 = np.stack([O.synth_adi(10, 32, seed=10 + i)[0] for i in range(3)])
a4 = np.linspace(0, 70, 10)
fo = O.pca_4d(c4, a4, ncomp=2, full_output=True)
fr = ref.pca(c4, a4, ncomp=2, full_output=True, verbose=False)
for nm, a, b in zip(("frame", "pcs", "recon", "res", "res_der", "ifs"), fo, fr):
    if nm == "pcs":
        continue
    check("pca 4d %s" % nm, a, b, atol=1e-4)
    check("pca 4d %s dtype" % nm, [a.dtype.itemsize], [b.dtype.itemsize], exact=True)

cube, ang = O.synth_adi(30, 64, seed=8)
for kw in (dict(asize=8, ncomp=3, fwhm=4, delta_rot=(0.1, 1)),
           dict(asize=8, ncomp=2, fwhm=4, delta_rot=0.5, radius_int=4, max_frames_lib=12),
           dict(asize=10, ncomp=(1, 2, 3), fwhm=4, delta_rot=(0.1, 1), n_segments=2)):
    co, cd, fo_ = O.pca_annular(cube, ang, full_output=True, **kw)
    cr, cdr, fr_ = ref.pca_annular(cube, ang, full_output=True, verbose=False, nproc=1, **kw)
    check("pca_annular %s cube_out" % kw, co, cr, atol=1e-4)
    check("pca_annular %s cube_der" % kw, cd, cdr, atol=1e-4)
    check("pca_annular %s frame" % kw, fo_, fr_, atol=1e-4)

print()
print("FAILURES: %d %s" % (len(FAILS), FAILS))
sys.exit(1 if FAILS else 0)
