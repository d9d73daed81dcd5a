"""GPU parity tests, end to end: vip_amd.psfsub.pca / cube_derotate / cube_collapse through the C ABI
against fixtures frozen from the reference (tests/golden) and the CPU oracle; tolerances: residual
cubes and frames max|d| < 1e-4 on max|cube| ~ 10 data (BASELINE.json), index sets bit-exact."""
import os

import numpy as np
import pytest

from conftest import load_golden, sign_align
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.mark.parametrize("tag,kw", [("k3", dict(ncomp=3)), ("eigen", dict(ncomp=3, svd_mode="eigen")),
                                    ("tmean", dict(ncomp=2, scaling="temp-mean")),
                                    ("tstd", dict(ncomp=2, scaling="temp-standard")),
                                    ("smean", dict(ncomp=2, scaling="spat-mean")),
                                    ("sstd", dict(ncomp=2, scaling="spat-standard")),
                                    ("mask", dict(ncomp=3, mask_center_px=5)),
                                    ("mean", dict(ncomp=4, collapse="mean"))])
def test_pca_small_golden(tag, kw):
    from vip_amd.psfsub import pca
    g = load_golden("g6_pca_small")
    out = pca(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    assert len(out) == 5
    for nm, a in zip(("frame", "pcs", "recon", "res", "resder"), out):
        b = g["%s_%s" % (tag, nm)]
        assert a.shape == b.shape and a.dtype == b.dtype, nm
        if nm == "pcs":
            a = sign_align(a, b)
        assert np.abs(a - b).max() < (5e-4 if nm == "recon" else TOL), (tag, nm, np.abs(a - b).max())
    fr = pca(g["cube"], g["angles"], verbose=False, **kw)
    assert np.abs(fr - g[tag + "_frame"]).max() < TOL


def test_pca_c1_golden():
    """BASELINE.json configs[0]: 50x128x128, ncomp=5 (FFT derotation path, L=512)."""
    from vip_amd.psfsub import pca
    g = load_golden("g6_pca_c1")
    cube, ang = O.synth_adi(50, 128, seed=int(g["seed"]))
    fr, pcs, recon, res, resd = pca(cube, ang, ncomp=5, full_output=True, verbose=False)
    c0 = 64 - 8
    assert np.abs(fr - g["frame"]).max() < TOL
    assert np.abs(res[:, c0:c0 + 16, c0:c0 + 16] - g["res_crop"]).max() < TOL
    assert np.abs(resd[:, c0:c0 + 16, c0:c0 + 16] - g["resder_crop"]).max() < TOL
    assert np.abs(np.sum(res.astype(np.float64), axis=0) - g["res_sum"]).max() < 2e-3
    assert np.abs(np.sum(resd.astype(np.float64), axis=0) - g["resder_sum"]).max() < 2e-3


def test_pca_rdi_and_cevr_and_clamp(capsys):
    from vip_amd.psfsub import pca
    g = load_golden("g6_pca_small")
    fr = pca(g["cube"], g["angles"], cube_ref=g["cube_ref"], ncomp=3, verbose=False)
    assert np.abs(fr - g["rdi_frame"]).max() < TOL
    fo = O.pca_fullframe(g["cube"], g["angles"], ncomp=3, cube_ref=g["cube_ref"], full_output=True)
    out = pca(g["cube"], g["angles"], cube_ref=g["cube_ref"], ncomp=3, verbose=False, full_output=True)
    for nm, a, b in zip(("frame", "pcs", "recon", "res", "resder"), out, fo):
        if nm == "pcs":
            a = sign_align(a, b)
        assert np.abs(a - b).max() < (5e-4 if nm == "recon" else TOL), nm
    # float ncomp (CEVR)
    fr = pca(g["cube"], g["angles"], ncomp=0.9, verbose=False)
    assert np.abs(fr - O.pca_fullframe(g["cube"], g["angles"], ncomp=0.9)).max() < TOL
    # ncomp > n is clamped with a message, not an error
    fr = pca(g["cube"], g["angles"], ncomp=40, verbose=False)
    assert "Number of PCs too high" in capsys.readouterr().out
    assert np.abs(fr).max() < 1e-3
    with pytest.raises(ValueError):
        pca(g["cube"], g["angles"], ncomp=0, verbose=False)
    with pytest.raises(ValueError):
        pca(g["cube"], g["angles"][:-1], ncomp=2, verbose=False)


def test_pca_4d_golden():
    from vip_amd.psfsub import pca
    g = load_golden("g6_pca_4d")
    out = pca(g["cube"], g["angles"], ncomp=2, full_output=True, verbose=False)
    assert len(out) == 6
    frame, pcs, recon, res, resd, ifs = out
    assert frame.dtype == np.float64 and ifs.dtype == np.float64 and res.dtype == np.float32
    assert pcs.shape == (3, 2, 32, 32) and recon.shape == g["cube"].shape
    assert np.abs(frame - g["frame"]).max() < TOL
    assert np.abs(res - g["res"]).max() < TOL
    assert np.abs(resd - g["resder"]).max() < TOL
    assert np.abs(ifs - g["ifs"]).max() < TOL
    assert np.abs(pca(g["cube"], g["angles"], ncomp=2, verbose=False) - g["frame"]).max() < TOL
    # frame-only calls take the batched path (one Gram / eigensolver / derotation launch for all channels): it must
    # agree with the per-channel loop that full_output=True runs
    for kw in (dict(scaling="temp-mean"), dict(mask_center_px=3, collapse="mean"), dict(collapse_ifs="median", ncomp=3),
               dict(scaling="spat-standard", collapse="sum")):
        kw = dict(dict(ncomp=2), **kw)
        loop = pca(g["cube"], g["angles"], full_output=True, verbose=False, **kw)[0]
        fast = pca(g["cube"], g["angles"], verbose=False, **kw)
        assert fast.dtype == np.float64 and np.abs(fast - loop).max() < 2e-5, kw


def test_raw_c_abi_binding_as_in_integration_md():
    """The ctypes stub of INTEGRATION.md section 3, verbatim in spirit: dlopen, vipmi_create on the current stream, one
    vipmi_pca_fullframe_f32 call with plain pointers and sizes, vipmi_last_error on failure -- no vip_amd front end."""
    import ctypes
    import os
    import torch
    from conftest import ROOT
    g = load_golden("g6_pca_small")
    cube_np, angle_list, ncomp = g["cube"], g["angles"], 3
    n, N = cube_np.shape[0], cube_np.shape[1]
    lib = ctypes.CDLL(os.path.join(ROOT, "vip_amd", "libvipmi.so"))
    lib.vipmi_last_error.restype = ctypes.c_char_p
    ctx = ctypes.c_void_p()
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.vipmi_create(0, ctypes.c_void_p(stream), ctypes.byref(ctx)) == 0
    cube = torch.from_numpy(cube_np).cuda()
    frame = torch.empty((N, N), dtype=torch.float32, device="cuda")
    angles = np.ascontiguousarray(O.check_pa_vector(angle_list), dtype=np.float64)
    lib.vipmi_pca_fullframe_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + \
        [ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5
    st = lib.vipmi_pca_fullframe_f32(ctx, cube.data_ptr(), angles.ctypes.data, n, N, ncomp, 0, None, 0,
                                     frame.data_ptr(), None, None, None, None)
    assert st == 0, lib.vipmi_last_error().decode()
    torch.cuda.synchronize()
    assert np.abs(frame.cpu().numpy() - g["k3_frame"]).max() < TOL
    # error convention: negative status + message, no exception / abort
    st = lib.vipmi_pca_fullframe_f32(ctx, cube.data_ptr(), angles.ctypes.data, n, N, 0, 0, None, 0,
                                     frame.data_ptr(), None, None, None, None)
    assert st < 0 and b"PCs" in lib.vipmi_last_error()
    lib.vipmi_destroy.argtypes = [ctypes.c_void_p]
    assert lib.vipmi_destroy(ctx) == 0


def test_device_tensor_api_and_algo_params():
    import torch
    from vip_amd.psfsub import pca, PCA_Params
    g = load_golden("g6_pca_small")
    ct = torch.from_numpy(g["cube"]).cuda()
    fr = pca(ct, g["angles"], ncomp=3, verbose=False)
    assert isinstance(fr, torch.Tensor) and fr.is_cuda
    assert np.abs(fr.cpu().numpy() - g["k3_frame"]).max() < TOL
    params = PCA_Params(cube=g["cube"], angle_list=g["angles"], ncomp=3, verbose=False)
    fr2 = pca(algo_params=params)
    assert np.abs(fr2 - g["k3_frame"]).max() < TOL
    fr3 = pca(g["cube"].astype(np.float64), g["angles"], ncomp=3, verbose=False)
    assert fr3.dtype == np.float64


def test_svd_wrapper_golden():
    from vip_amd.psfsub.svd import svd_wrapper
    g = load_golden("g1_svd")
    for tag in ("a", "b"):
        M = g["M_" + tag]
        for mode in ("lapack", "eigen"):
            V = svd_wrapper(M, mode, 6, False)
            Vr = g["V_%s_%s" % (mode, tag)]
            assert V.shape == Vr.shape and V.dtype == np.float32
            assert np.abs(sign_align(V, Vr) - Vr).max() < 3e-5
            U, S, V2 = svd_wrapper(M, mode, 6, False, full_output=True)
            np.testing.assert_allclose(S, g["S_%s_%s" % (mode, tag)], rtol=2e-5)
            assert U.shape == ((M.shape[0], 6) if mode == "lapack" else (6, M.shape[0]))
    # reference tests/pre_3_10/test_pca_svd.py:10-20
    mat = np.random.RandomState(42).randn(20, 100)
    U, S, V = svd_wrapper(mat, "lapack", 20, False, full_output=True)
    assert np.allclose(np.abs(U @ np.diag(S) @ V), np.abs(mat), atol=1e-2)


@pytest.mark.parametrize("seed", [0, 1])
def test_pca_vs_oracle_noise_and_structured(seed):
    """Pure-noise cube (flat spectrum) and the structured generator at a non power-of-two size."""
    from vip_amd.psfsub import pca
    rng = np.random.default_rng(seed)
    cube = rng.standard_normal((24, 45, 45)).astype(np.float32)
    ang = np.linspace(-20, 200, 24)
    fr = pca(cube, ang, ncomp=4, verbose=False)
    assert np.abs(fr - O.pca_fullframe(cube, ang, ncomp=4)).max() < TOL
    cube, ang = O.synth_adi(20, 101, seed=seed)
    out = pca(cube, ang, ncomp=5, verbose=False, full_output=True)
    ref = O.pca_fullframe(cube, ang, ncomp=5, full_output=True)
    assert np.abs(out[0] - ref[0]).max() < TOL
    assert np.abs(out[3] - ref[3]).max() < TOL
    assert np.abs(out[4] - ref[4]).max() < TOL


@pytest.mark.parametrize("tag,kw", [("a", dict(asize=8, ncomp=3, fwhm=4, delta_rot=(0.1, 1))),
                                    ("b", dict(asize=8, ncomp=2, fwhm=4, delta_rot=0.5, radius_int=4, max_frames_lib=12)),
                                    ("c", dict(asize=10, ncomp=(1, 2, 3), fwhm=4, delta_rot=(0.1, 1), n_segments=2))])
def test_pca_annular_golden(tag, kw):
    from vip_amd.psfsub import pca_annular
    g = load_golden("g6_pca_annular")
    cube_out, cube_der, frame = pca_annular(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    assert cube_out.shape == g["cube"].shape and cube_out.dtype == np.float32
    assert np.abs(cube_out - g[tag + "_cube_out"]).max() < TOL
    assert np.abs(frame - g[tag + "_frame"]).max() < TOL
    fr = pca_annular(g["cube"], g["angles"], verbose=False, **kw)
    assert np.abs(fr - g[tag + "_frame"]).max() < TOL


@pytest.mark.parametrize("tag,kw", [("a", dict(ncomp=[1, 3, 6], asize=8, fwhm=4, delta_rot=(0.1, 1))),
                                    ("b", dict(ncomp=[2, 4], asize=5, fwhm=4, delta_rot=0.5, n_segments=2, radius_int=5,
                                               scaling="temp-standard", collapse="mean"))])
def test_pca_annular_list_ncomp_golden(tag, kw):
    """list ncomp (reference pca_local.py:665-668,892-902) against the reference's outputs; each entry also equals
    the single-ncomp run."""
    from vip_amd.psfsub import pca_annular
    g = load_golden("g12_annular_list")
    co, cd, frames = pca_annular(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    assert isinstance(frames, list) and len(frames) == len(kw["ncomp"])
    assert [str(co.dtype), str(cd.dtype), str(frames[0].dtype)] == list(g[tag + "_dtypes"])
    assert co.shape == g[tag + "_out"].shape and cd.shape == g[tag + "_der"].shape
    assert np.abs(co - g[tag + "_out"]).max() < TOL
    assert np.nanmax(np.abs(cd - g[tag + "_der"])) < TOL
    assert np.abs(np.stack(frames) - g[tag + "_frames"]).max() < TOL
    only = pca_annular(g["cube"], g["angles"], verbose=False, **kw)
    assert isinstance(only, list) and np.abs(np.stack(only) - g[tag + "_frames"]).max() < TOL
    kw1 = dict(kw, ncomp=kw["ncomp"][1])
    single = pca_annular(g["cube"], g["angles"], verbose=False, **kw1)
    assert np.abs(single - frames[1]).max() < 2e-5


@pytest.mark.parametrize("tag,kw", [("ref", dict(ref=True, ncomp=3, asize=8, fwhm=4, delta_rot=(0.1, 1))),
                                    ("sig", dict(sig=True, ncomp=2, asize=8, fwhm=4, delta_rot=0.5, scaling="temp-mean")),
                                    ("both", dict(ref=True, sig=True, ncomp=[2, 5], asize=10, fwhm=4, delta_rot=1,
                                                  n_segments=2))])
def test_pca_annular_ref_sig_golden(tag, kw):
    """pca_annular with cube_ref (RDI) / cube_sig against the reference's outputs (pca_local.py:716-724,862-891)."""
    from vip_amd.psfsub import pca_annular
    g = load_golden("g13_annular_ref_sig")
    kw = dict(kw)
    if kw.pop("ref", False):
        kw["cube_ref"] = g["cube_ref"]
    if kw.pop("sig", False):
        kw["cube_sig"] = g["cube_sig"]
    co, cd, fr = pca_annular(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    fr = np.stack(fr) if isinstance(fr, list) else fr
    assert co.shape == g[tag + "_out"].shape
    assert np.abs(co - g[tag + "_out"]).max() < TOL
    assert np.nanmax(np.abs(cd - g[tag + "_der"])) < TOL
    assert np.abs(fr - g[tag + "_frame"]).max() < TOL
    with pytest.raises(TypeError):
        pca_annular(g["cube"], g["angles"], cube_sig=g["cube_sig"][:-1], ncomp=2, asize=8, verbose=False)


def test_pca_annular_4d_with_reference_cube():
    """4-D cube + 4-D reference cube (pca_local.py:279-325): every channel equals the 3-D RDI call on that channel."""
    from vip_amd.psfsub import pca_annular
    c4 = np.stack([O.synth_adi(12, 40, seed=40 + i)[0] for i in range(3)])
    r4 = np.stack([O.synth_adi(6, 40, seed=60 + i)[0] for i in range(3)])
    ang = np.linspace(0, 80, 12)
    kw = dict(asize=8, ncomp=2, fwhm=4, delta_rot=(0.1, 1), verbose=False)
    co, cd, fr = pca_annular(c4, ang, cube_ref=r4, full_output=True, **kw)
    per = [pca_annular(c4[ch], ang, cube_ref=r4[ch], full_output=True, **kw) for ch in range(3)]
    assert np.abs(co - np.stack([p[0] for p in per])).max() < 1e-6
    assert np.abs(fr - np.mean(np.stack([p[2] for p in per]).astype(np.float64), axis=0)).max() < 1e-6
    ref0 = O.pca_annular(c4[0], ang, cube_ref=r4[0], asize=8, ncomp=2, fwhm=4, delta_rot=(0.1, 1), full_output=True)
    assert np.abs(per[0][0] - ref0[0]).max() < TOL


def test_rare_switches_golden():
    """n_segments='auto' (radian quirk kept), theta_init, radius_int, max_frames_lib; weighted-mean collapse -- against
    the reference's outputs."""
    from vip_amd.psfsub import pca, pca_annular
    g = load_golden("g19_switches")
    cube, ang, w = g["cube"], g["angles"], g["weights"]
    co, cd, fr = pca_annular(cube, ang, ncomp=2, asize=6, fwhm=4, delta_rot=(0.1, 0.8), n_segments="auto", theta_init=30,
                             radius_int=6, max_frames_lib=9, full_output=True, verbose=False)
    assert np.abs(co - g["ann_out"]).max() < TOL
    assert np.nanmax(np.abs(cd - g["ann_der"])) < TOL
    assert np.abs(fr - g["ann_frame"]).max() < TOL
    assert np.abs(pca(cube, ang, ncomp=3, collapse="wmean", weights=w, verbose=False) - g["pca_wmean"]).max() < TOL
    assert np.abs(pca_annular(cube, ang, ncomp=2, asize=8, fwhm=4, collapse="wmean", weights=w, verbose=False)
                  - g["ann_wmean"]).max() < TOL


def test_pca_annular_scaling_and_errors():
    from vip_amd.psfsub import pca_annular
    cube, ang = O.synth_adi(20, 48, seed=4)
    for sc in ("temp-mean", "spat-standard"):
        got = pca_annular(cube, ang, asize=6, ncomp=2, fwhm=4, delta_rot=0.3, scaling=sc, verbose=False, full_output=True)
        ref = O.pca_annular(cube, ang, asize=6, ncomp=2, fwhm=4, delta_rot=0.3, scaling=sc, full_output=True)
        assert np.abs(got[0] - ref[0]).max() < TOL, sc
        assert np.abs(got[2] - ref[2]).max() < TOL, sc
    with pytest.raises(RuntimeError):
        pca_annular(cube, np.linspace(0, 1, 20), asize=6, ncomp=2, fwhm=4, delta_rot=5.0, verbose=False)
    with pytest.raises(TypeError):
        pca_annular(cube, ang[:-1], asize=6, ncomp=2, verbose=False)


# ---- grid of PCs and PA-threshold frame rejection ------------------------------------------------

@pytest.mark.parametrize("tag,kw", [("grid_a", dict(ncomp=(1, 5))),
                                    ("grid_b", dict(ncomp=(2, 9, 3), scaling="temp-mean", mask_center_px=4)),
                                    ("grid_c", dict(ncomp=[1, 4, 6], collapse="mean")),
                                    ("grid_d", dict(ncomp=(1, 4), rdi=True))])
def test_pca_grid_golden(tag, kw):
    from vip_amd.psfsub import pca
    g = load_golden("g7_grid_rejection")
    kw = dict(kw)
    if kw.pop("rdi", False):
        kw["cube_ref"] = g["cube_ref"]
    out = pca(g["cube"], g["angles"], verbose=False, **kw)
    exp = g[tag + "_frames"]
    assert out.shape == exp.shape and out.dtype == exp.dtype
    assert np.abs(out - exp).max() < TOL
    fo = pca(g["cube"], g["angles"], verbose=False, full_output=True, **kw)
    assert len(fo) == 2 and np.abs(fo[0] - exp).max() < TOL
    if tag == "grid_a":
        assert list(fo[1]) == list(g["grid_a_pcs"])
        med = pca(g["cube"], g["angles"], verbose=False, med_of_npcs=True, **kw)
        assert np.abs(med - np.median(exp, axis=0)).max() < TOL


@pytest.mark.parametrize("tag,kw", [("rej_a", dict(ncomp=3, source_xy=(34, 24), fwhm=4, delta_rot=1, min_frames_pca=4)),
                                    ("rej_b", dict(ncomp=2, source_xy=(30, 30), fwhm=4, delta_rot=0.5, min_frames_pca=3,
                                                   max_frames_pca=8, scaling="temp-standard")),
                                    ("rej_c", dict(ncomp=2, source_xy=(10, 24), fwhm=5, delta_rot=1, min_frames_pca=2,
                                                   mask_center_px=3))])
def test_pca_pa_rejection_golden(tag, kw):
    from vip_amd.psfsub import pca
    g = load_golden("g7_grid_rejection")
    out = pca(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    assert len(out) == 4
    for nm, a in zip(("frame", "recon", "res", "resder"), out):
        b = g["%s_%s" % (tag, nm)]
        assert a.shape == b.shape and a.dtype == b.dtype, nm
        assert np.abs(a - b).max() < (5e-4 if nm == "recon" else TOL), (tag, nm, np.abs(a - b).max())
    fr = pca(g["cube"], g["angles"], verbose=False, **kw)
    assert np.abs(fr - g[tag + "_frame"]).max() < TOL


@pytest.mark.parametrize("tag,kw", [("plain", dict(ncomp=3)), ("scaled", dict(ncomp=2, scaling="temp-mean", mask_center_px=3)),
                                    ("rdi", dict(ncomp=3, rdi=True))])
def test_cube_sig_golden(tag, kw):
    """cube_sig (reference pca_fullfr.py:1652-1662,1717-1731) against the reference's outputs."""
    from vip_amd.psfsub import pca
    g = load_golden("g11_cube_sig")
    kw = dict(kw)
    if kw.pop("rdi", False):
        kw["cube_ref"] = g["cube_ref"]
    out = pca(g["cube"], g["angles"], cube_sig=g["cube_sig"], full_output=True, verbose=False, **kw)
    assert len(out) == 5
    for nm, a in zip(("frame", "pcs", "recon", "res", "resder"), out):
        b = g["%s_%s" % (tag, nm)]
        assert a.shape == b.shape and a.dtype == b.dtype, nm
        if nm == "pcs":
            a = sign_align(a, b)
        assert np.abs(a - b).max() < TOL, (tag, nm, np.abs(a - b).max())
    fr = pca(g["cube"], g["angles"], cube_sig=g["cube_sig"], verbose=False, **kw)
    assert np.abs(fr - g[tag + "_frame"]).max() < TOL


def test_cube_sig_pa_rejection_golden():
    from vip_amd.psfsub import pca
    g = load_golden("g11_cube_sig")
    N = g["cube"].shape[1]
    out = pca(g["cube"], g["angles"], ncomp=2, cube_sig=g["cube_sig"], source_xy=(N // 2 + 9, N // 2), fwhm=4,
              delta_rot=1, min_frames_pca=3, full_output=True, verbose=False)
    for nm, a in zip(("frame", "recon", "res", "resder"), out):
        b = g["sxy_%s" % nm]
        assert a.shape == b.shape and a.dtype == b.dtype, nm
        assert np.abs(a - b).max() < (5e-4 if nm == "recon" else TOL), (nm, np.abs(a - b).max())
    with pytest.raises(TypeError):
        pca(g["cube"], g["angles"], ncomp=2, cube_sig=g["cube_sig"][:-1], verbose=False)
    with pytest.raises(NotImplementedError):
        pca(g["cube"], g["angles"], ncomp=(1, 3), cube_sig=g["cube_sig"], verbose=False)


def test_pca_pa_rejection_errors():
    from vip_amd.psfsub import pca
    g = load_golden("g7_grid_rejection")
    with pytest.raises(TypeError):
        pca(g["cube"], g["angles"], ncomp=2, source_xy=(34, 24), verbose=False)            # fwhm / delta_rot missing
    with pytest.raises(RuntimeError):
        pca(g["cube"], g["angles"], ncomp=2, source_xy=(26, 24), fwhm=4, delta_rot=20, verbose=False)   # empty libraries
    fr = pca(g["cube"], g["angles"], ncomp=(1, 3), source_xy=(34, 24), fwhm=4, verbose=False)   # S/N-scored grid: a frame
    assert fr.shape == g["cube"].shape[1:]


def test_sharded_modes_world1_use_the_device_path():
    """vip_amd.dist with its DEFAULT compute (the device pca) on one rank: survey mode and the 4-D channel split."""
    from vip_amd import dist as D
    from vip_amd.psfsub import pca
    cubes = [O.synth_adi(10, 32, seed=s)[0] for s in (1, 2, 3)]
    angs = [np.linspace(0, 60, 10)] * 3
    frames = D.pca_cubes(cubes, angs, ncomp=2, verbose=False).cpu().numpy()
    for i in range(3):
        assert np.abs(frames[i] - pca(cubes[i], angs[i], ncomp=2, verbose=False)).max() < 1e-6
    g = load_golden("g6_pca_4d")
    frame, ifs = D.pca_4d(g["cube"], g["angles"], ncomp=2, verbose=False)
    assert np.abs(frame - g["frame"]).max() < TOL and np.abs(ifs - g["ifs"]).max() < TOL


def test_single_cube_sharded_path_world1():
    """vip_amd.dist.pca_single_cube with the device kernels (one rank: the slab / exchange code paths degenerate to
    local copies) reproduces pca()."""
    from vip_amd import dist as D
    from vip_amd.psfsub import pca
    cube, ang = O.synth_adi(20, 64, seed=4)
    ref = pca(cube, ang, ncomp=4, verbose=False)
    got = D.pca_single_cube(cube, O.check_pa_vector(ang), 4).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-5
    got_mean = D.pca_single_cube(cube, O.check_pa_vector(ang), 4, collapse="mean").cpu().numpy()
    assert np.abs(got_mean - pca(cube, ang, ncomp=4, collapse="mean", verbose=False)).max() < 1e-5


# ---- median subtraction and STIM maps ------------------------------------------------------------

@pytest.mark.parametrize("tag,kw", [("a", dict()), ("b", dict(radius_int=4, collapse="mean")),
                                    ("c", dict(rdi=True, collapse_ref="median")), ("d", dict(rdi=True, collapse_ref="mean"))])
def test_median_sub_golden(tag, kw):
    from vip_amd.psfsub import median_sub
    g = load_golden("g8_medsub_stim")
    kw = dict(kw)
    if kw.pop("rdi", False):
        kw["cube_ref"] = g["cube_ref"]
    co, cd, fr = median_sub(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    for nm, a, b in (("out", co, g["ms_%s_out" % tag]), ("der", cd, g["ms_%s_der" % tag]), ("frame", fr, g["ms_%s_frame" % tag])):
        assert a.shape == b.shape and a.dtype == b.dtype, nm
        assert np.abs(a - b).max() < TOL, (tag, nm, np.abs(a - b).max())
    assert np.abs(median_sub(g["cube"], g["angles"], verbose=False, **kw) - g["ms_%s_frame" % tag]).max() < TOL
    with pytest.raises(NotImplementedError):
        median_sub(g["cube"], g["angles"], mode="annular", nframes=None, verbose=False)


@pytest.mark.parametrize("tag,kw", [("a", dict(asize=4, fwhm=4, delta_rot=1)),
                                    ("b", dict(asize=6, fwhm=3, delta_rot=0.5, nframes=2, collapse="mean")),
                                    ("c", dict(asize=5, fwhm=4, radius_int=4, nframes=6)),
                                    ("d", dict(asize=4, rdi=True, collapse_ref="mean"))])
def test_median_sub_annular_golden(tag, kw):
    """median_sub(mode='annular') (reference medsub.py:316-371,602-676) against the reference's outputs."""
    from vip_amd.psfsub import median_sub
    g = load_golden("g14_medsub_annular")
    kw = dict(kw)
    if kw.pop("rdi", False):
        kw["cube_ref"] = g["cube_ref"]
    co, cd, fr = median_sub(g["cube"], g["angles"], mode="annular", full_output=True, verbose=False, **kw)
    for got, nm in ((co, "out"), (cd, "der"), (fr, "frame")):
        exp = g["ms_%s_%s" % (tag, nm)]
        assert got.shape == exp.shape and got.dtype == exp.dtype, nm
        assert np.nanmax(np.abs(got - exp)) < (2e-6 if nm == "out" else 2e-5), (tag, nm)
    with pytest.raises(TypeError):
        median_sub(g["cube"], g["angles"], mode="annular", nframes=3, verbose=False)
    with pytest.raises(RuntimeError):
        median_sub(g["cube"], g["angles"], mode="nope", verbose=False)


@pytest.mark.parametrize("tag,kw", [("ff", dict(radius_int=5)), ("ff_mean", dict(radius_int=3, collapse="mean")),
                                    ("ann", dict(mode="annular", asize=4, fwhm=4, radius_int=4, nframes=4))])
def test_median_sub_odd_frame_count_keeps_its_exact_zeros(tag, kw):
    """g29 (round 6, round-5 ADVICE): with an odd frame count the median IS one of the samples, `cube - median` is exactly 0 for one
    frame per pixel, and the reference's mask_val = 0 rotation (radius_int > 0) resets exactly those pixels.  The projection's zero
    guard (project.hip keep_nonzero, option sub_guard) must not touch this subtraction."""
    from vip_amd.psfsub import median_sub
    g = load_golden("g29_medsub_odd")
    co, cd, fr = median_sub(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    exp = g["ms_%s_out" % tag]
    assert co.shape == exp.shape and co.dtype == exp.dtype
    assert np.array_equal(co == 0, exp == 0), "the exact zeros of cube - median differ from the reference's"
    assert np.nanmax(np.abs(co - exp)) < 2e-6
    assert np.nanmax(np.abs(cd - g["ms_%s_der" % tag])) < 2e-5, np.nanmax(np.abs(cd - g["ms_%s_der" % tag]))
    assert np.nanmax(np.abs(fr - g["ms_%s_frame" % tag])) < 2e-5
    # and the guard is back on for the PCA calls of the same context
    from vip_amd import backend as B
    assert int(B.get_context().get_option("sub_guard")) == 1


def test_stim_maps_golden():
    from vip_amd.metrics import stim_map, inverse_stim_map, normalized_stim_map
    g = load_golden("g8_medsub_stim")
    s = stim_map(g["resder"])
    assert s.shape == g["stim"].shape and np.abs(s - g["stim"]).max() < 1e-4
    # the inverse / normalised maps divide by a per-pixel standard deviation: compare relative to the map's scale
    for got, exp in ((inverse_stim_map(g["res"], g["angles"]), g["stim_inv"]),
                     (normalized_stim_map(g["res"], g["angles"]), g["stim_norm"]),
                     (normalized_stim_map(g["res"], g["angles"], mask=5), g["stim_norm_mask"])):
        assert got.shape == exp.shape
        assert np.abs(got - exp).max() < 2e-3 * max(1.0, np.abs(exp).max())


def test_pca_many_matches_serial():
    """Independent cubes issued through two streams in asynchronous mode give exactly the serial results."""
    from vip_amd.psfsub import pca, pca_many
    cubes, angs = zip(*[O.synth_adi(20 + 2 * i, 64, seed=30 + i) for i in range(5)])
    serial = [pca(c, a, ncomp=3, verbose=False) for c, a in zip(cubes, angs)]
    many = pca_many(list(cubes), list(angs), depth=2, ncomp=3)
    assert len(many) == 5
    for s_, m in zip(serial, many):
        assert m.dtype == s_.dtype and np.array_equal(s_, m)
    # a failing call must not leave the library in asynchronous mode
    with pytest.raises(ValueError):
        pca_many(list(cubes), list(angs), ncomp=0)
    from vip_amd import backend
    assert backend.is_async() is False


def test_pca_many_keeps_float64_cubes_on_the_float64_route():
    """pca_many(cubes) == [pca(c, a) ...] also for float64 cubes of detector counts (round-5 ADVICE): the uploader must not round
    them to float32 (golden g28: that alone costs 2e-3 on the frame)."""
    from vip_amd.psfsub import pca, pca_many
    g = load_golden("g28_f64_counts")
    cubes = [g["cube"], g["cube"] + 3.0, g["cube"][:30].copy()]
    angs = [g["angles"], g["angles"], g["angles"][:30]]
    serial = [pca(c, a, ncomp=4, verbose=False) for c, a in zip(cubes, angs)]
    many = pca_many(cubes, angs, depth=2, ncomp=4)
    for s_, m in zip(serial, many):
        assert m.dtype == np.float64 and np.array_equal(s_, m, equal_nan=True)
    assert np.nanmax(np.abs(many[0] - g["frame64_k4"])) < 1e-4
    import torch
    many_dev = pca_many([torch.from_numpy(c).cuda() for c in cubes], angs, depth=2, ncomp=4)
    assert np.array_equal(many_dev[0].cpu().numpy().astype(np.float64), many[0], equal_nan=True)


@pytest.mark.parametrize("n,N,k", [(50, 128, 5), (120, 256, 8)])
def test_pipelined_calls_are_bit_identical(n, N, k):
    """Many pca() calls in flight on two streams (asynchronous mode: dynamic task queues, the barrier-free task ring of
    shear 1, the gate between calls): every frame must be bit-identical to a serial call's."""
    import torch
    from vip_amd import backend as B
    from vip_amd.psfsub import pca
    cube, ang = O.synth_adi(n, N, seed=9)
    ct = torch.from_numpy(cube).cuda()
    ref = pca(ct, ang, ncomp=k, verbose=False).clone()
    streams = [torch.cuda.Stream() for _ in range(2)]
    B.set_async(True)
    try:
        outs = []
        for i in range(60):
            with torch.cuda.stream(streams[i % 2]):
                outs.append(pca(ct, ang, ncomp=k, verbose=False))
        torch.cuda.synchronize()
        B.check_deferred()
    finally:
        B.set_async(False)
    assert all(torch.equal(o, ref) for o in outs)


def test_pca_annular_4d_golden():
    """4-D cube without scale_list: per-channel annular PCA, then the spectral collapse (pca_local.py:279-325)."""
    from vip_amd.psfsub import pca_annular
    g = load_golden("g9_annular_4d")
    co, cd, fr = pca_annular(g["cube"], g["angles"], asize=8, ncomp=2, fwhm=4, delta_rot=(0.1, 1), full_output=True,
                             verbose=False)
    assert co.shape == g["cube_out"].shape and cd.shape == g["cube_der"].shape and fr.shape == g["frame"].shape
    assert fr.dtype == g["frame"].dtype
    assert np.abs(co - g["cube_out"]).max() < TOL
    assert np.abs(cd - g["cube_der"]).max() < TOL
    assert np.abs(fr - g["frame"]).max() < TOL
    assert np.abs(pca_annular(g["cube"], g["angles"], asize=8, ncomp=[2, 2, 2], fwhm=4, delta_rot=(0.1, 1),
                              verbose=False) - g["frame"]).max() < TOL


# ---- ADI+mSDI -------------------------------------------------------------------------------------

def test_rescaling_wavelengths_golden():
    from vip_amd.preproc.rescaling import cube_rescaling_wavelengths
    g = load_golden("g10_msdi")
    r = cube_rescaling_wavelengths(g["cube"][:, 0], g["scale_list"])
    assert r[0].shape == g["scw_cube"].shape and r[2:4] == (40, 40)
    assert np.abs(r[0] - g["scw_cube"]).max() < 2e-5
    assert np.abs(r[1] - g["scw_frame"]).max() < 2e-5
    ri = cube_rescaling_wavelengths(g["scw_cube"], g["scale_list"], full_output=True, inverse=True, y_in=32, x_in=32,
                                    collapse="mean")
    assert ri[0].shape == g["scw_inv_cube"].shape
    assert np.abs(ri[0] - g["scw_inv_cube"]).max() < 2e-5
    assert np.abs(ri[1] - g["scw_inv_frame"]).max() < 2e-5
    fr = cube_rescaling_wavelengths(g["scw_cube"], g["scale_list"], full_output=False, inverse=True, y_in=32, x_in=32,
                                    collapse="mean")
    assert np.abs(fr - g["scw_inv_frame"]).max() < 2e-5


@pytest.mark.parametrize("tag,kw", [("d22", dict(ncomp=(2, 2))), ("dN2", dict(ncomp=(None, 2))), ("d2N", dict(ncomp=(2, None))),
                                    ("dmask", dict(ncomp=(2, 2), mask_center_px=3, scaling="temp-mean", collapse_ifs="median")),
                                    ("drange", dict(ncomp=(2, 2), ifs_collapse_range=(1, 4)))])
def test_msdi_double_golden(tag, kw):
    from vip_amd.psfsub import pca
    g = load_golden("g10_msdi")
    out = pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="double", full_output=True, verbose=False, **kw)
    assert len(out) == 3
    for nm, a in zip(("frame", "rcc", "rcc_der"), out):
        b = g["%s_%s" % (tag, nm)]
        assert a.shape == b.shape and a.dtype == b.dtype, (nm, a.dtype, b.dtype)
        assert np.abs(a - b).max() < TOL, (tag, nm, np.abs(a - b).max())
    fr = pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="double", verbose=False, **kw)
    assert np.abs(fr - g[tag + "_frame"]).max() < TOL


@pytest.mark.parametrize("tag,kw", [("s3", dict(ncomp=3)), ("s3nocrop", dict(ncomp=3, crop_ifs=False)),
                                    ("s2mask", dict(ncomp=2, mask_center_px=3, scaling="temp-standard"))])
def test_msdi_single_golden(tag, kw):
    from vip_amd.psfsub import pca
    g = load_golden("g10_msdi")
    out = pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="single", full_output=True, verbose=False, **kw)
    assert len(out) == 4
    for nm, a in zip(("frame", "allfr", "desc", "adi"), out):
        b = g["%s_%s" % (tag, nm)]
        assert a.shape == b.shape, (nm, a.shape, b.shape)
        assert np.abs(a - b).max() < TOL, (tag, nm, np.abs(a - b).max())


def test_msdi_single_grid_and_reference_cube_golden():
    """single-pass ADI+mSDI: tuple / list ``ncomp`` (grid of frames), S/N-scored grid at ``source_xy``, 4-D ``cube_ref``
    (RSDI and ARSDI) against the reference's own outputs (g23) resp. the oracle"""
    from vip_amd.psfsub import pca
    g = load_golden("g23_msdi_single_more")
    c4, a4, sc = g["cube"], g["angles"], g["scale_list"]
    kw = dict(scale_list=sc, adimsdi="single", verbose=False)
    fr = pca(c4, a4, ncomp=(1, 4), **kw)
    assert fr.shape == g["grid_frames"].shape and np.abs(fr - g["grid_frames"]).max() < TOL
    fr, pcs = pca(c4, a4, ncomp=[2, 5], full_output=True, **kw)
    assert list(pcs) == list(g["list_pcs"]) and np.abs(fr - g["list_frames"]).max() < TOL
    fr = pca(c4, a4, ncomp=(1, 5, 2), ifs_collapse_range=(1, 4), collapse="mean", scaling="temp-mean", mask_center_px=3,
             **kw)
    assert np.abs(fr - g["grid_range_median"]).max() < TOL
    out = pca(c4, a4, ncomp=3, cube_ref=g["cube_ref"], full_output=True, **kw)
    for nm, a in zip(("frame", "allfr", "desc", "adi"), out):
        assert a.shape == g["ref_" + nm].shape and np.abs(a - g["ref_" + nm]).max() < TOL, nm
    # ARSDI: the science frames join the library (pca_fullfr.py:503-508) -- against the oracle
    both = np.concatenate((c4, g["cube_ref"]), axis=1)
    fr = pca(c4, a4, ncomp=3, cube_ref=g["cube_ref"], ref_strategy="ARSDI", **kw)
    assert np.abs(fr - O.pca_adimsdi_single(c4, a4, sc, 3, cube_ref=both)).max() < TOL
    # S/N-scored grid: table and best frame (the S/N itself is host code shared with the 3-D grid)
    cubeout, best, table = pca(c4, a4, ncomp=(1, 4), source_xy=(20.0, 12.0), fwhm=4.0, full_output=True, **kw)
    assert np.abs(cubeout - g["grid_frames"]).max() < TOL
    k = int(np.argmax(table["S/Ns"].values))
    assert np.array_equal(best, cubeout[k]) and list(table["PCs"]) == [1, 2, 3, 4]
    with pytest.raises(NotImplementedError):             # (still outside the accelerated path; cube_ref / source_xy: G26)
        pca(c4, a4, ncomp=(2, 2), scale_list=sc, adimsdi="double", smooth_first_pass=2, verbose=False)


@pytest.mark.parametrize("N", [64, 65])
def test_msdi_larger_sizes_golden(N):
    """ADI+mSDI with 7 channels at an even and an odd frame size against the reference's outputs."""
    from vip_amd.psfsub import pca
    g = load_golden("g18_msdi_%d" % N)
    fo = pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="double", ncomp=(2, 3), full_output=True,
             verbose=False)
    for nm, a in zip(("frame", "rcc", "rcc_der"), fo):
        b = g["d_%s" % nm]
        assert a.shape == b.shape and a.dtype == b.dtype, nm
        assert np.nanmax(np.abs(a - b)) < TOL, (nm, np.nanmax(np.abs(a - b)))
    fs = pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="single", ncomp=4, full_output=True,
             verbose=False)
    assert np.abs(fs[0] - g["s_frame"]).max() < TOL
    assert np.nanmax(np.abs(fs[3] - g["s_adi"])) < TOL


def test_msdi_errors():
    from vip_amd.psfsub import pca
    g = load_golden("g10_msdi")
    with pytest.raises(TypeError):
        pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="double", ncomp=2, verbose=False)
    with pytest.raises(ValueError):
        pca(g["cube"], g["angles"], scale_list=g["scale_list"][:3], adimsdi="double", ncomp=(1, 1), verbose=False)


def test_left_eigv_golden():
    """pca(left_eigv=True) against the reference (g24): same frame / cubes as the standard projection, pcs = temporal
    modes (k x n); svd_wrapper(left_eigv=True) -> (n x k); the reference's incompatibilities raise"""
    from vip_amd.psfsub import pca
    from vip_amd.psfsub.svd import svd_wrapper
    g = load_golden("g24_left_eigv")
    cube, ang = g["cube"], g["angles"]
    out = pca(cube, ang, ncomp=3, left_eigv=True, full_output=True, verbose=False)
    assert len(out) == 5
    for nm, a in zip(("frame", "pcs", "recon", "res", "resd"), out):
        b = g["left_" + nm]
        assert a.shape == b.shape, (nm, a.shape, b.shape)
        if nm == "pcs":
            a = a * np.sign(np.sum(a * b, axis=1, keepdims=True))
        assert np.nanmax(np.abs(a - b)) < TOL, nm
    fr = pca(cube, ang, ncomp=3, left_eigv=True, verbose=False)
    assert np.abs(fr - g["left_frame_only"]).max() < TOL
    M = cube.reshape(14, -1)
    for mode, key in (("lapack", "svd_left_lapack"), ("arpack", "svd_left_arpack")):
        U = svd_wrapper(M, mode, 4, False, left_eigv=True)
        assert U.shape == (14, 4) and np.abs(np.abs(U.T @ g[key]) - np.eye(4)).max() < 1e-5
    with pytest.raises(NotImplementedError):
        pca(cube, ang, ncomp=3, left_eigv=True, cube_ref=cube, verbose=False)


def test_more_than_64_components():
    """ncomp > 64: full-frame PCA (fused entry and svd_wrapper) through the looping matrix-in-L2 eigensolver"""
    from vip_amd.psfsub import pca
    from vip_amd.psfsub.svd import svd_wrapper
    n, N, k = 160, 40, 100
    cube, ang = O.synth_adi(n, N, seed=11)
    got = pca(cube, ang, ncomp=k, verbose=False)
    assert np.abs(got - O.pca_fullframe(cube, ang, ncomp=k)).max() < TOL
    M = cube.reshape(n, -1).astype(np.float64)
    V = svd_wrapper(M.astype(np.float32), "lapack", k, False)
    Vr = O.svd_wrapper(M, "lapack", k)
    # same row space (PCs of the noise floor are defined up to rotations inside near-degenerate groups)
    assert np.abs(V.T @ (V @ M.T) - Vr.T @ (Vr @ M.T)).max() < 1e-3 * np.abs(M).max()


def test_annular_staged_eigensolve_matches_the_per_segment_route(monkeypatch):
    """With 512+ libraries in a call the annular path solves the libraries of ALL segments in one batched launch
    (Gram / sub-Gram stage, one eigensolve, coefficient / residual stage); below that, and with VIPMI_ANNULAR_STAGED=0,
    segment by segment.  Same frames from both (list ncomp, reference cube and two segments per annulus included), and
    the float64 restatement of the reference agrees."""
    from vip_amd.psfsub import pca_annular
    cube, _ = O.synth_adi(130, 56, seed=11)
    ang = np.linspace(0, 160, 130)
    ref_cube, _ = O.synth_adi(6, 56, seed=12)
    for kw in (dict(asize=6, ncomp=3, fwhm=4, delta_rot=(0.1, 1)),
               dict(asize=8, ncomp=[1, 4], fwhm=4, delta_rot=0.5, n_segments=2, max_frames_lib=60),
               dict(asize=6, ncomp=(1, 2, 3, 2), fwhm=4, delta_rot=1, cube_ref=ref_cube, scaling="temp-mean")):
        monkeypatch.setenv("VIPMI_ANNULAR_STAGED", "1")
        a = pca_annular(cube, ang, full_output=True, verbose=False, **kw)
        monkeypatch.setenv("VIPMI_ANNULAR_STAGED", "0")
        b = pca_annular(cube, ang, full_output=True, verbose=False, **kw)
        ref = O.pca_annular(cube, ang, full_output=True, **kw)
        for x, y, z in zip(a, b, ref):
            x, y, z = (np.stack(t) if isinstance(t, list) else t for t in (x, y, z))
            assert np.array_equal(np.isfinite(x), np.isfinite(y))
            assert np.nanmax(np.abs(x - y)) < 2e-5
            assert np.nanmax(np.abs(x - z)) < TOL


@pytest.mark.parametrize("tag,kw", [("a", dict(asize=8, ncomp=3, fwhm=4, delta_rot=(0.1, 1))),
                                    ("b", dict(asize=8, ncomp=2, fwhm=4, delta_rot=0.5, radius_int=4, max_frames_lib=12)),
                                    ("c", dict(asize=10, ncomp=(1, 2, 3), fwhm=4, delta_rot=(0.1, 1), n_segments=2))])
def test_annular_fused_front_golden(tag, kw, monkeypatch):
    """Round 6: the fronts of all segments in a handful of launches (one gather, one ragged int8 Gram product, one eigensolve, one
    coefficient launch, one residual product that scatters through the pixel list; csrc/annular.hip annular_gram_all_f32 /
    annular_apply_all_f32).  Forced on the small goldens of the reference: the one-pixel overlap of the last two annuli (the later
    segment wins), per-annulus ncomp, radius_int and two segments per annulus included; and against the per-segment route."""
    from vip_amd.psfsub import pca_annular
    g = load_golden("g6_pca_annular")
    monkeypatch.setenv("VIPMI_ANNULAR_FUSED", "1")
    cube_out, cube_der, frame = pca_annular(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    monkeypatch.setenv("VIPMI_ANNULAR_FUSED", "0")
    co0, cd0, fr0 = pca_annular(g["cube"], g["angles"], full_output=True, verbose=False, **kw)
    assert cube_out.shape == g["cube"].shape and cube_out.dtype == np.float32
    assert np.abs(cube_out - g[tag + "_cube_out"]).max() < TOL
    assert np.abs(frame - g[tag + "_frame"]).max() < TOL
    assert np.array_equal(cube_out == 0, co0 == 0)                 # the same pixels written, the rest untouched
    assert np.abs(cube_out - co0).max() < 5e-6 and np.nanmax(np.abs(frame - fr0)) < 5e-6


@pytest.mark.parametrize("scaling", [None, "temp-mean", "temp-standard"])
def test_annular_fused_front_matches_the_per_segment_route(scaling, monkeypatch):
    """The fused front at a size where it is the default (130 frames x 96 px would not be: forced), with temporal scalings (applied to
    the whole gathered matrix at once), a tuple ncomp and several segments per annulus, against the per-segment route and the oracle."""
    from vip_amd.psfsub import pca_annular
    cube, _ = O.synth_adi(130, 96, seed=61)
    ang = np.linspace(0, 150, 130)
    kw = dict(asize=12, ncomp=(2, 3, 4, 3), fwhm=4, delta_rot=(0.2, 1), n_segments=[1, 2, 3, 2], max_frames_lib=70, scaling=scaling)
    monkeypatch.setenv("VIPMI_ANNULAR_FUSED", "1")
    a = pca_annular(cube, ang, full_output=True, verbose=False, **kw)
    monkeypatch.setenv("VIPMI_ANNULAR_FUSED", "0")
    b = pca_annular(cube, ang, full_output=True, verbose=False, **kw)
    ref = O.pca_annular(cube, ang, full_output=True, **kw)
    for x, y, z in zip(a, b, ref):
        assert np.array_equal(np.isfinite(x), np.isfinite(y))
        assert np.nanmax(np.abs(x - y)) < 5e-6, np.nanmax(np.abs(x - y))
        assert np.nanmax(np.abs(x - z)) < TOL


def test_annular_libraries_beyond_512_frames():
    """PCA libraries of more than 512 frames per annulus (max_frames_lib raised far above the reference's default 200)
    leave the batched eigensolver for the matrix-in-L2 one, library after library (zero-padded sub-Gram matrices, no
    active-size argument).  That route is checked (a) against the float64 restatement of the reference with the switch
    lowered to libraries of 128+ frames on a cube the CPU finishes in seconds, (b) at its real size for sanity against
    the batched route on slightly smaller libraries."""
    from vip_amd import backend as B
    from vip_amd.psfsub import pca_annular
    ctx = B.get_context()
    cube, _ = O.synth_adi(170, 32, seed=3)
    ang = np.linspace(0, 200, 170)
    kw = dict(asize=8, ncomp=3, fwhm=4, delta_rot=(0.1, 1), max_frames_lib=160)
    ref = O.pca_annular(cube, ang, **kw)
    ctx.set_option("ann_large_min", 127)
    try:
        fr = pca_annular(cube, ang, verbose=False, **kw)
    finally:
        ctx.set_option("ann_large_min", 512)
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(fr), ok)
    assert np.abs(fr[ok] - ref[ok]).max() < 2e-4, np.abs(fr[ok] - ref[ok]).max()
    cube, _ = O.synth_adi(560, 32, seed=2)
    ang = np.linspace(0, 300, 560)
    kw = dict(asize=8, ncomp=3, fwhm=4, delta_rot=(0.1, 1))
    big = pca_annular(cube, ang, max_frames_lib=540, verbose=False, **kw)
    small = pca_annular(cube, ang, max_frames_lib=512, verbose=False, **kw)
    ok = np.isfinite(small)
    assert np.array_equal(np.isfinite(big), ok) and ok.sum() > 100
    # (different libraries, so not the same frame: the same speckle floor, and nothing blown up by the padded solves)
    assert 0.2 < np.abs(big[ok]).std() / np.abs(small[ok]).std() < 5.0


def test_full_output_for_numpy_callers_lands_in_owned_pinned_blocks():
    """Round 6: cube-sized results (full_output) are copied straight into pinned host memory and handed over as numpy arrays that own
    their block (backend.to_host_many): bit-identical to the device tensors, untouched by later calls while the caller holds them,
    float64 in -> float64 out converted on the device."""
    import gc
    import torch
    from vip_amd import backend as B
    from vip_amd.psfsub import pca, pca_annular
    cube, ang = O.synth_adi(40, 256, seed=5)                    # 10.5 MB per cube-sized array: above the pinned threshold
    dev = pca(torch.from_numpy(cube).cuda(), ang, ncomp=4, full_output=True, verbose=False)
    out1 = pca(cube, ang, ncomp=4, full_output=True, verbose=False)
    keep = [a.copy() for a in out1]
    for a, d in zip(out1, dev):
        assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.writeable
        assert np.array_equal(a, d.cpu().numpy(), equal_nan=True)
    held = B._pin_out["bytes"]
    assert held >= 3 * cube.nbytes
    out2 = pca(cube * 0.5, ang, ncomp=4, full_output=True, verbose=False)          # a second call must not touch the first results
    for a, k_ in zip(out1, keep):
        assert np.array_equal(a, k_, equal_nan=True)
    assert not np.array_equal(out2[3], out1[3])
    del out1, out2, a
    gc.collect()
    assert B._pin_out["bytes"] < held                          # blocks handed back when the arrays die
    c64 = cube.astype(np.float64)
    o64 = pca(c64, ang, ncomp=4, full_output=True, verbose=False)
    assert all(a.dtype == np.float64 for a in o64) and np.nanmax(np.abs(o64[0] - keep[0])) < 1e-4
    co, cd, fr = pca_annular(cube, ang, ncomp=2, asize=32, fwhm=4, full_output=True, verbose=False)
    cod, cdd, frd = pca_annular(torch.from_numpy(cube).cuda(), ang, ncomp=2, asize=32, fwhm=4, full_output=True, verbose=False)
    assert np.array_equal(co, cod.cpu().numpy()) and np.array_equal(cd, cdd.cpu().numpy(), equal_nan=True)


def test_more_than_6144_frames():
    """Beyond 6144 frames the three n-vectors of the exact leading-k solver no longer fit the LDS.  The leading pairs still come
    from the verified fast path when the spectrum allows (csrc/eigh_chfsi.hip); since round 6 everything else -- a fast path that
    gives up, the whole spectrum of the eigen family's full output, CEVR / a float ncomp -- is served by the same exact solver with
    its vectors in global memory (eigh_tri_large.hip, tri_xl_kernel<32, true>; slow, behind a RuntimeWarning) instead of raising.
    No library call anywhere."""
    import warnings
    from vip_amd import backend as B
    from vip_amd.psfsub import pca
    n, N, k = B.MAX_EIGH_LDS_N + 56, 16, 6
    cube, _ = O.synth_adi(n, N, seed=3)
    ang = np.linspace(0, 170, n)
    ref = O.pca_fullframe(cube, ang, ncomp=k)
    got = pca(cube, ang, ncomp=k, verbose=False)
    assert np.abs(got - ref).max() < TOL
    ctx = B.get_context()
    assert ctx.get_option("eigh_fast_last_reason") == 0 and ctx.get_option("eigh_fast_last_locked") == k
    from vip_amd.psfsub.svd import svd_wrapper
    mat = cube.reshape(n, -1)
    U, S, V = svd_wrapper(mat, "lapack", 4, verbose=False, full_output=True)
    Ur, Sr, Vr = O.svd_wrapper(mat, "lapack", 4, full_output=True)
    assert U.shape == (n, 4) and S.shape == (4,) and V.shape == (4, N * N)
    np.testing.assert_allclose(S, Sr, rtol=2e-5)
    assert np.abs(sign_align(V, Vr) - Vr).max() < TOL
    # the eigen family returns the WHOLE spectrum (svd.py:454-462): the exact solver, vectors in global memory, with a warning
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        Ue, Se, Ve = svd_wrapper(mat, "eigen", 4, verbose=False, full_output=True)
    assert any("vectors in global memory" in str(w_.message) for w_ in rec)
    Uo, So, Vo = O.svd_wrapper(mat, "eigen", 4, full_output=True)
    assert Se.shape == So.shape
    np.testing.assert_allclose(Se[:N * N - 1], So[:N * N - 1], rtol=5e-5, atol=1e-3 * So[0])
    assert np.abs(sign_align(Ve, Vo) - Vo).max() < TOL
    # the fast path switched off (as if it had given up): same frame from the exact path
    for c in B.all_contexts():
        c.set_option("eigh_fast", 0)
    try:
        got2 = pca(cube, ang, ncomp=k, verbose=False)
        U2, S2, V2 = svd_wrapper(mat, "lapack", 4, verbose=False, full_output=True)
    finally:
        for c in B.all_contexts():
            c.set_option("eigh_fast", 1)
    assert np.abs(got2 - ref).max() < TOL
    np.testing.assert_allclose(S2, Sr, rtol=2e-5)
    assert np.abs(sign_align(V2, Vr) - Vr).max() < TOL


def test_8192_frames_whole_spectrum_and_cevr():
    """Round-5 VERDICT #7: svd_wrapper(M, 'lapack', k, full_output=True), the eigen family's whole spectrum and a float ncomp (CEVR,
    svd.py:216-339) on an 8192-frame cube of 64 x 64 px against numpy."""
    import warnings
    from vip_amd.psfsub import pca
    from vip_amd.psfsub.svd import svd_wrapper
    n, N, k = 8192, 64, 5
    rng = np.random.default_rng(8192)
    base, _ = O.synth_adi(256, N, seed=81)
    cube = (base[rng.integers(0, 256, n)] + 0.3 * rng.standard_normal((n, N, N))).astype(np.float32)
    ang = np.linspace(0, 120, n)
    mat = cube.reshape(n, -1)
    _, w, Vt = np.linalg.svd(mat.astype(np.float64), full_matrices=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U, S, V = svd_wrapper(mat, "lapack", k, verbose=False, full_output=True)
        np.testing.assert_allclose(S, w[:k], rtol=2e-5)
        assert U.shape == (n, k) and V.shape == (k, N * N)
        assert np.abs(V @ V.T - np.eye(k)).max() < 1e-4
        # the leading row space against numpy's
        assert np.linalg.norm(V @ Vt[k:].T, 2) < 1e-3
        Ue, Se, Ve = svd_wrapper(mat, "eigen", k, verbose=False, full_output=True)
        m = min(n, N * N)
        np.testing.assert_allclose(Se[:m - 1], w[:m - 1], rtol=1e-4, atol=1e-4 * w[0])
        # float ncomp: the number of components from the cumulative explained variance ratio
        fr = pca(cube, ang, ncomp=0.5, verbose=False)
    # the number of components numpy's spectrum gives for that CEVR (svd.py:253-257,335), and the frame of the integer call with it
    # (the integer path at this size is pinned against the oracle in test_more_than_6144_frames: the oracle's own derotation of 8192
    # frames would be a minute of CPU here)
    exp_var = w ** 2 / (w.shape[0] - 1)
    kc = int(np.searchsorted(np.cumsum(exp_var / exp_var.sum()), 0.5) + 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = pca(cube, ang, ncomp=kc, verbose=False)
    assert fr.shape == (N, N) and np.array_equal(fr, ref, equal_nan=True), kc
    # and the residuals of that call against numpy's own components (reference pca_fullfr.py:1727-1731)
    m64 = mat.astype(np.float64)
    res = (m64 - (m64 @ Vt[:kc].T) @ Vt[:kc]).reshape(n, N, N)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = pca(cube, ang, ncomp=kc, full_output=True, verbose=False)[3]
    assert np.abs(got - res).max() < TOL


def test_more_than_2048_frames():
    """2048 < n <= 6144 frames: the 3-vector variant of the matrix-in-L2 eigensolver (eigh_tri_large.hip, tri_xl_kernel),
    fused entry, RDI reference library and svd_wrapper: ADI cube and RDI reference library of 2100 frames"""
    from vip_amd.psfsub import pca
    n, N, k = 2100, 32, 10
    cube, _ = O.synth_adi(n, N, seed=n)
    ang = np.linspace(0, 170, n)
    got = pca(cube, ang, ncomp=k, verbose=False)
    assert np.abs(got - O.pca_fullframe(cube, ang, ncomp=k)).max() < TOL
    small, a2 = O.synth_adi(30, N, seed=6)
    got = pca(small, a2, ncomp=5, cube_ref=cube, verbose=False)
    assert np.abs(got - O.pca_fullframe(small, a2, ncomp=5, cube_ref=cube)).max() < TOL
    from vip_amd.psfsub.svd import svd_wrapper
    V = svd_wrapper(cube.reshape(n, -1)[:, :900], "lapack", 4, verbose=False)
    Vr = O.svd_wrapper(cube.reshape(n, -1)[:, :900], "lapack", 4)
    assert np.abs(sign_align(V, Vr) - Vr).max() < TOL


def test_pca_with_opencv_style_rotation():
    """pca(..., imlib='opencv'): same residuals as the parity path, derotated with the interpolating warp (oracle:
    project_subtract + warp_rotate + median); full-frame and annular."""
    from vip_amd.psfsub import pca, pca_annular
    n, N, k = 24, 49, 4
    cube, ang = O.synth_adi(n, N, seed=77)
    out = pca(cube, ang, ncomp=k, imlib="opencv", interpolation="bicubic", full_output=True, verbose=False)
    frame, res, resd = out[0], out[3], out[4]
    ref_res = O.pca_fullframe(cube, ang, ncomp=k, full_output=True)[3]
    assert np.abs(res - ref_res).max() < TOL
    ref_der = np.stack([O.warp_rotate(ref_res[i], -ang[i], "bicubic") for i in range(n)])
    assert np.abs(resd - ref_der).max() < TOL
    assert np.abs(frame - np.median(ref_der, axis=0)).max() < TOL
    co, cd, fa = pca_annular(cube, ang, ncomp=3, asize=6, fwhm=4, imlib="opencv", interpolation="lanczos4",
                             full_output=True, verbose=False)
    co_fft = pca_annular(cube, ang, ncomp=3, asize=6, fwhm=4, full_output=True, verbose=False)[0]
    assert np.array_equal(co, co_fft)                       # the residuals do not depend on the rotation
    ref_der = np.stack([O.warp_rotate(co[i], -ang[i], "lanczos4") for i in range(n)])
    assert np.abs(cd - ref_der).max() < 1e-5 and np.abs(fa - np.median(ref_der, axis=0)).max() < 1e-5


def test_median_sub_and_stim_with_opencv_style_rotation():
    """median_sub / STIM maps with imlib='opencv': model subtraction as on the parity path, derotation by the warp
    (oracle: warp_rotate), then the reference's collapse / STIM formula (metrics/stim.py:20-58)."""
    from vip_amd.psfsub import median_sub
    from vip_amd.metrics import inverse_stim_map
    n, N = 18, 41
    cube, ang = O.synth_adi(n, N, seed=91)
    co, cd, fr = median_sub(cube, ang, imlib="opencv", interpolation="bilinear", full_output=True, verbose=False)
    ref_out = cube - np.median(cube, axis=0)
    assert np.abs(co - ref_out).max() < 1e-5
    ref_der = np.stack([O.warp_rotate(ref_out[i], -ang[i], "bilinear") for i in range(n)])
    assert np.abs(cd - ref_der).max() < 1e-5 and np.abs(fr - np.median(ref_der, axis=0)).max() < 1e-5
    inv = inverse_stim_map(ref_out, ang, imlib="opencv", interpolation="bicubic")
    d = np.stack([O.warp_rotate(ref_out[i], ang[i], "bicubic") for i in range(n)]).astype(np.float64)
    exp = O.stim_map(d)
    assert np.abs(inv - exp).max() < 2e-3 * max(1.0, np.abs(exp).max())


def test_pca_4d_with_opencv_style_rotation():
    """4-D cube without scale_list (per-channel ADI, pca_fullfr.py:544-658) under imlib='opencv': every channel is the
    3-D result; ADI+mSDI keeps requiring 'vip-fft'."""
    from vip_amd.psfsub import pca
    nch, n, N = 3, 14, 33
    cubes = np.stack([O.synth_adi(n, N, seed=200 + c)[0] for c in range(nch)])
    ang = O.synth_adi(n, N, seed=200)[1]
    kw = dict(ncomp=2, imlib="opencv", interpolation="bilinear", verbose=False)
    got = pca(cubes, ang, **kw)
    per_channel = np.stack([pca(cubes[c], ang, **kw) for c in range(nch)])
    assert got.shape == (N, N) and np.abs(got - per_channel.mean(axis=0)).max() < 1e-5
    with pytest.raises(NotImplementedError):
        pca(cubes, ang, scale_list=np.linspace(1.0, 1.2, nch), adimsdi="single", ncomp=1, imlib="opencv", verbose=False)


def test_trimmean_collapse_through_the_pipelines():
    """collapse='trimmean' through pca / pca_annular / median_sub: the reference calls cube_collapse(mode=collapse) with its
    default n=50 (subsampling.py:30,88-96) -- on short cubes the python-slice semantics of sorted[k:k+n] apply."""
    from vip_amd.psfsub import pca, pca_annular, median_sub
    for n in (24, 70):                                    # n < 50 (slice clipped) and n > 50
        cube, ang = O.synth_adi(n, 64, seed=4)
        fr = pca(cube, ang, ncomp=3, collapse="trimmean", verbose=False)
        ref = O.pca_fullframe(cube, ang, ncomp=3, collapse="trimmean")
        assert np.abs(fr - ref).max() < TOL
        fo = pca(cube, ang, ncomp=3, collapse="trimmean", verbose=False, full_output=True)
        assert np.abs(fo[0] - ref).max() < TOL
        fa = pca_annular(cube, ang, ncomp=2, asize=8, fwhm=4, collapse="trimmean", verbose=False)
        ra = O.pca_annular(cube, ang, ncomp=2, asize=8, fwhm=4, collapse="trimmean")
        assert np.abs(fa - ra).max() < TOL
        fm = median_sub(cube, ang, collapse="trimmean", verbose=False)
        assert np.abs(fm - O.median_sub_fullfr(cube, ang, collapse="trimmean")).max() < TOL


def test_stim_is_not_a_public_collapse_mode():
    from vip_amd.psfsub import pca, pca_annular
    cube, ang = O.synth_adi(12, 40, seed=2)
    with pytest.raises(TypeError):
        pca(cube, ang, ncomp=2, collapse="stim", verbose=False)
    with pytest.raises(TypeError):
        pca_annular(cube, ang, ncomp=2, asize=8, collapse="stim", verbose=False)


def test_pca_4d_frame_rejection_and_grid_return_packing():
    """4-D cube with source_xy (per-channel frame rejection) and with a list ncomp applied as a grid to every channel
    (reference pca_fullfr.py:603-658,772-788): tuple layouts and values against the per-channel oracle."""
    from vip_amd.psfsub import pca
    chans = [O.synth_adi(20, 48, seed=s)[0] for s in (1, 2, 3)]
    cube4 = np.stack(chans)
    ang = np.linspace(0, 90, 20)
    kw = dict(ncomp=2, source_xy=(34, 24), fwhm=4, delta_rot=0.5, min_frames_pca=3)
    out = pca(cube4, ang, full_output=True, verbose=False, **kw)
    assert len(out) == 5                                   # frame, recon_cube, residuals, residuals_, ifs_adi_frames
    per = [O.pca_pa_rejection(c, ang, 2, (34, 24), 4, 0.5, min_frames_pca=3, full_output=True) for c in chans]
    ifs = np.stack([p[0] for p in per])
    assert out[4].shape == (3, 48, 48) and np.abs(out[4] - ifs).max() < TOL
    assert np.abs(out[0] - ifs.mean(axis=0)).max() < TOL
    assert out[1].shape == (3, 20, 48, 48)
    fr = pca(cube4, ang, verbose=False, **kw)
    assert np.abs(fr - ifs.mean(axis=0)).max() < TOL
    # grid: list of 2 PCs on 3 channels -> (n_grid, y, x) collapsed over channels
    grid = pca(cube4, ang, ncomp=[1, 3], verbose=False)
    exp = np.stack([np.mean([O.pca_fullframe(c, ang, ncomp=k) for c in chans], axis=0) for k in (1, 3)])
    assert grid.shape == (2, 48, 48) and np.abs(grid - exp).max() < TOL
    g3 = pca(cube4, ang, ncomp=[1, 3], verbose=False, full_output=True)
    assert len(g3) == 3 and g3[1] == [[1, 3]] * 3 and g3[2].shape == (3, 2, 48, 48)


def test_sharded_annular_world1_matches_pca_annular():
    """vip_amd.dist.pca_annular (annuli -> residual columns -> frame shards -> derotation -> row slabs -> collapse) on one
    rank with the device kernels against pca_annular itself."""
    from vip_amd import dist as D
    from vip_amd.psfsub import pca_annular
    cube, ang = O.synth_adi(24, 64, seed=6)
    for kw in (dict(ncomp=3, asize=8, fwhm=4), dict(ncomp=2, asize=8, fwhm=4, radius_int=6, n_segments=2)):
        ref = pca_annular(cube, ang, verbose=False, **kw)
        got = D.pca_annular(cube, ang, **kw).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.nanmax(np.abs(got - ref)) < 1e-5


def test_bench_sharded_modes_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --mode ...` spawns its own two ranks (no torchrun on the command line); here they share
    this GPU and use gloo for the collectives (VIPMI_BENCH_DEVICE / VIPMI_BENCH_BACKEND): the multi-rank code paths that
    RCCL runs on an 8-GPU node -- ragged all_gather / all_to_all lists included."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, VIPMI_BENCH_BACKEND="gloo", VIPMI_BENCH_DEVICE="0")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k_, None)
    for mode, extra in (("single-cube", ["--frames", "30", "--size", "128", "--ncomp", "4"]),
                        ("annular", ["--frames", "30", "--size", "128"])):
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                             "--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", mode] + extra,
                            capture_output=True, text=True, env=env, timeout=600)
        assert cp.returncode == 0, cp.stderr[-3000:]
        rec = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
        assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0


def test_bench_survey_mode_two_ranks_with_strong_legs_on_one_gpu():
    """The default (survey) mode at --gpus 2, self-spawned, small cube: one JSON line, n_gpus = 2, weak scaling; the
    strong-scaling legs only run at the BASELINE shape, so a second run checks `strong` at 400 x 512 x 512 with one
    rank (selfcheck of the three sharded routines against the single-GPU calls included)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, VIPMI_BENCH_BACKEND="gloo", VIPMI_BENCH_DEVICE="0")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k_, None)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                         "--frames", "40", "--size", "128", "--ncomp", "5", "--no-cpu-baseline"],
                        capture_output=True, text=True, env=env, timeout=600)
    assert cp.returncode == 0, cp.stderr[-3000:]
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["cubes_per_step"] == 2
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                         "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert cp.returncode == 0, cp.stderr[-3000:]
    rec = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    st = rec["strong"]
    assert "error" not in st, st
    assert max(st["selfcheck_max_abs_diff_vs_one_gpu"].values()) < 1e-4
    for key in ("single_cube_c5", "annular_c3", "ifs_4d_c4"):
        assert st[key]["value"] > 0
    assert rec["sustained"]["seconds"] > 1.0 and rec["value_serial"] > 0


def test_bench_refuses_more_gpus_than_the_box_has():
    """`bench.py --gpus N` must never silently run fewer ranks (round-2 VERDICT): more ranks than devices is an error,
    and so is a WORLD_SIZE that contradicts --gpus."""
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VIPMI_BENCH_DEVICE")}
    want = torch.cuda.device_count() + 1
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1"],
                        capture_output=True, text=True, env=env, timeout=300)
    assert cp.returncode != 0 and "GPU(s) visible" in cp.stderr and "{" not in cp.stdout
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                        capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert cp.returncode != 0 and "WORLD_SIZE" in cp.stderr


def test_rccl_world_2_when_two_devices_are_visible():
    """Two ranks on two GPUs over RCCL (tests/helpers/rccl_world2.py): the three torch.distributed partitions and the
    C entry vipmi_pca_fullframe_sharded_f32 against pca().  Skipped on a one-GPU box."""
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cp = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29541",
                         os.path.join(ROOT, "tests", "helpers", "rccl_world2.py")],
                        capture_output=True, text=True, env=env, timeout=900)
    assert cp.returncode == 0 and "OK" in cp.stdout, (cp.stdout[-2000:], cp.stderr[-3000:])
    for ln in cp.stdout.splitlines():
        if ln.startswith("case"):
            assert float(ln.split()[-1]) < 1e-5, ln


def test_c_client_of_the_c_abi(tmp_path):
    """The boundary is a C ABI: a plain C program (tests/helpers/cabi_client.c: gcc, the HIP runtime for the device
    buffers, libvipmi.so -- no Python, no PyTorch in that process) runs vipmi_pca_fullframe_f32 and must write the frame
    that pca() returns."""
    import shutil
    import subprocess
    from conftest import ROOT
    from vip_amd.psfsub import pca
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler on this box")
    exe = str(tmp_path / "cabi_client")
    libdir = os.path.join(ROOT, "vip_amd")
    cp = subprocess.run(["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                         os.path.join(ROOT, "tests", "helpers", "cabi_client.c"), "-o", exe, "-L", libdir, "-lvipmi",
                         "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr[-2000:]
    n, N, k = 30, 64, 4
    cube, ang = O.synth_adi(n, N, seed=17)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([n, N, k], dtype=np.int64).tofile(f)
        cube.astype(np.float32).tofile(f)
        np.asarray(ang, dtype=np.float64).tofile(f)
    cp = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, (cp.stdout, cp.stderr[-2000:])
    got = np.fromfile(fout, dtype=np.float32).reshape(N, N)
    ref = pca(cube, ang, ncomp=k, verbose=False)
    assert np.array_equal(got, ref, equal_nan=True)
    assert np.nanmax(np.abs(got - O.pca_fullframe(cube, ang, ncomp=k))) < TOL


def test_sharded_c_entry_on_an_rccl_communicator_world_1():
    """include/vipmi.h: vipmi_rccl_* + vipmi_pca_fullframe_sharded_f32 (RCCL resolved at run time, communicator created by
    the library, all-reduce / grouped send-recv exchanges incl. the sends to self) against pca() -- one rank, which is
    what one GPU allows; the partition logic for more ranks is the one the gloo tests run through dist.pca_single_cube"""
    import subprocess
    import sys
    from conftest import ROOT
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_world1.py")], capture_output=True,
                        text=True, timeout=300)
    assert cp.returncode == 0 and "OK" in cp.stdout, (cp.stdout[-2000:], cp.stderr[-3000:])
    for ln in cp.stdout.splitlines():
        if ln.startswith("case"):
            assert float(ln.split()[-1]) < 1e-6, ln


# ---- SURVEY 8(f) #1 / #4 remainders: pca_annulus, the S/N-scored PCA grid, a PPPCA-shaped caller ---------------------

def test_pca_annulus_golden():
    """reference psfsub/utils_pca.py:617-755 (G20: ADI, scaling, RDI, residual cubes, 4-D)."""
    from vip_amd.psfsub import pca_annulus
    g = load_golden("g20_pca_annulus")
    cube, ang, cref = g["cube"], g["angles"], g["cube_ref"]
    fr = pca_annulus(cube, ang, ncomp=3, annulus_width=8, r_guess=14)
    assert fr.dtype == np.float32 and np.nanmax(np.abs(fr - g["adi"])) < TOL
    assert np.array_equal(np.isnan(fr), np.isnan(g["adi"]))
    fr = pca_annulus(cube, ang, ncomp=2, annulus_width=6, r_guess=10, scaling="temp-mean", collapse="mean")
    assert np.nanmax(np.abs(fr - g["adi_mean_tm"])) < TOL
    fr = pca_annulus(cube, ang, ncomp=4, annulus_width=8, r_guess=13.5, cube_ref=cref)
    assert np.nanmax(np.abs(fr - g["rdi"])) < TOL
    cd = pca_annulus(cube, ang, ncomp=3, annulus_width=8, r_guess=14, collapse=None)
    assert cd.shape == cube.shape and np.nanmax(np.abs(cd - g["cube_res_der"])) < TOL
    cr = pca_annulus(cube, None, ncomp=3, annulus_width=8, r_guess=14, collapse=None)
    assert np.nanmax(np.abs(cr - g["cube_res"])) < TOL
    ifs = pca_annulus(g["cube4"], ang, ncomp=[2, 3, 2], annulus_width=8, r_guess=14, collapse="median", collapse_ifs="mean")
    assert ifs.dtype == np.float64 and np.nanmax(np.abs(ifs - g["ifs"])) < TOL
    with pytest.raises(ValueError):
        pca_annulus(g["cube4"], ang, ncomp=2, annulus_width=8, r_guess=14, collapse=None)
    with pytest.raises(TypeError):
        pca_annulus(g["cube4"], ang, ncomp=[2, 3], annulus_width=8, r_guess=14)


def _cube_with_companion(n=20, N=48, seed=9, r=11.0, flux=3.0):
    cube, ang = O.synth_adi(n, N, seed=seed, planet=False)
    ang = np.linspace(0, 80, n)
    yy, xx = np.mgrid[:N, :N]
    c = N // 2
    for i, th in enumerate(np.deg2rad(ang)):                    # rotates with the field: fixed position after derotation
        py, px = c + r * np.sin(th + 0.3), c + r * np.cos(th + 0.3)
        cube[i] += (flux * np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2 * 1.7 ** 2))).astype(np.float32)
    return cube, ang, (c + r * np.cos(0.3), c + r * np.sin(0.3))


def test_pca_grid_scored_by_snr():
    """pca(ncomp=<tuple>, source_xy=..., fwhm=...): grid of frames on the device + S/N scoring on the host
    (reference psfsub/utils_pca.py:239-277,364-402, pca_fullfr.py:706-713,779-790), against the oracle."""
    from vip_amd.psfsub import pca, pca_grid
    cube, ang, (sx, sy) = _cube_with_companion()
    xy = (float(round(sx)), float(round(sy)))
    ref = O.pca_grid_snr(cube, ang, (1, 6, 1), xy, 4.0)
    out = pca(cube, ang, ncomp=(1, 6, 1), source_xy=xy, fwhm=4.0, full_output=True, verbose=False)
    assert len(out) == 3
    cubeout, frame, table = out
    assert cubeout.shape == (6, 48, 48) and np.nanmax(np.abs(cubeout - ref[0])) < TOL
    assert list(table["PCs"]) == ref[2]
    assert np.abs(np.array(table["S/Ns"], dtype=float) - np.array(ref[3], dtype=float)).max() < 2e-3 * max(1.0, np.max(np.abs(ref[3])))
    assert np.abs(np.array(table["fluxes"], dtype=float) - np.array(ref[4], dtype=float)).max() < 1e-3
    assert np.nanmax(np.abs(frame - ref[1])) < TOL                                 # the frame of the best S/N
    only = pca(cube, ang, ncomp=(1, 6, 1), source_xy=xy, fwhm=4.0, verbose=False)
    assert np.array_equal(only, frame, equal_nan=True)
    # the public pca_grid, every figure of merit + the annular mode
    for fm in ("px", "max", "mean"):
        co, fin, df, opt = pca_grid(cube, ang, fwhm=4.0, range_pcs=[2, 4], source_xy=xy, fmerit=fm, verbose=False)
        r2 = O.pca_grid_snr(cube, ang, [2, 4], xy, 4.0, fmerit=fm)
        assert opt == r2[5] and np.abs(np.array(df["S/Ns"], dtype=float) - np.array(r2[3], dtype=float)).max() < 2e-3 * max(1.0, np.max(np.abs(r2[3])))
    co, fin, df, opt = pca_grid(cube, ang, fwhm=4.0, range_pcs=(1, 3), source_xy=xy, mode="annular", annulus_width=8,
                                verbose=False)
    for i, k in enumerate((1, 2, 3)):
        exp = O.pca_annulus(cube, ang, k, 8, np.hypot(xy[0] - 24, xy[1] - 24))
        assert np.nanmax(np.abs(co[i] - exp)) < TOL
    with pytest.raises(ValueError):
        pca_grid(cube, ang, range_pcs=(1, 3), source_xy=xy, verbose=False)          # fwhm missing


def test_pppca_shaped_caller():
    """A caller shaped like the reference's PPPCA object (objects/pppca.py:131-417): a dataclass carrying the PCA_Params
    fields plus its own, calling ``pca(**{"algo_params": self, **rot_options})`` and unpacking the result by the tuple
    layouts of ``_find_pca_mode`` (:285-413) -- every layout the accelerated path serves."""
    from dataclasses import dataclass, field
    from vip_amd.psfsub import pca, PCA_Params

    @dataclass
    class PostProc(PCA_Params):
        dataset: object = None
        results: object = None
        extras: dict = field(default_factory=dict)

        def run(self, **rot_options):
            res = pca(**{"algo_params": self, **rot_options})
            self._find_pca_mode(res)
            return res

        def _find_pca_mode(self, res):
            it = isinstance(self.ncomp, (tuple, list))
            if self.scale_list is not None and self.adimsdi == "double":
                self.frame_final, self.cube_residuals, self.cube_residuals_der = res
            elif self.scale_list is not None and not it:
                self.frame_final, self.cube_residuals, _, _ = res                  # (:759-761 returns four elements)
            elif (self.cube_ref is not None or self.source_xy is None) and it:
                if self.cube.ndim == 4:
                    self.frames_final, self.pc_list, _ = res
                else:
                    self.frames_final, self.pc_list = res
            elif (self.cube_ref is not None or self.source_xy is None) and not it:
                if self.cube.ndim == 4:
                    (self.frame_final, self.pcs, self.cube_reconstructed, self.cube_residuals, self.cube_residuals_der,
                     _) = res
                else:
                    (self.frame_final, self.pcs, self.cube_reconstructed, self.cube_residuals,
                     self.cube_residuals_der) = res
            elif self.source_xy is not None and it:
                self.final_residuals_cube, self.frame_final, _ = res
            else:
                if self.cube.ndim == 4:
                    self.frame_final, self.cube_reconstructed, self.cube_residuals, self.cube_residuals_der, _ = res
                else:
                    self.frame_final, self.cube_reconstructed, self.cube_residuals, self.cube_residuals_der = res

    g = load_golden("g6_pca_small")
    cube, ang = g["cube"], g["angles"]
    common = dict(cube=cube, angle_list=ang, full_output=True, verbose=False)
    # ADI_FULLFRAME_STANDARD, 3-D
    pp = PostProc(ncomp=3, **common)
    pp.run()
    assert np.abs(pp.frame_final - O.pca_fullframe(cube, ang, ncomp=3)).max() < TOL
    assert pp.pcs.shape == (3,) + cube.shape[1:] and pp.cube_residuals_der.shape == cube.shape
    # ... with rot_options passed beside algo_params (mask_val = 0 -> zeros restored after the rotation)
    pp = PostProc(ncomp=3, mask_center_px=3, **common)
    pp.run(mask_val=0, interp_zeros=True, ker=1)
    assert np.abs(pp.frame_final - O.pca_fullframe(cube, ang, ncomp=3, mask_center_px=3)).max() < TOL
    # ADI_FULLFRAME_GRID, 3-D
    pp = PostProc(ncomp=(1, 3), **common)
    pp.run()
    assert pp.pc_list == [1, 2, 3] and np.abs(pp.frames_final - O.pca_grid_frames(cube, ang, (1, 3))).max() < TOL
    # PCA_ROT_THRESH, 3-D
    pp = PostProc(ncomp=2, source_xy=(31, 20), fwhm=4, delta_rot=0.3, min_frames_pca=2, **common)
    pp.run()
    exp = O.pca_pa_rejection(cube, ang, 2, (31, 20), 4, 0.3, min_frames_pca=2)
    assert np.nanmax(np.abs(pp.frame_final - exp)) < TOL and pp.cube_reconstructed.shape == cube.shape
    # PCA_GRID_SN, 3-D
    c2, a2, (sx, sy) = _cube_with_companion()
    pp = PostProc(cube=c2, angle_list=a2, ncomp=(1, 4), source_xy=(float(round(sx)), float(round(sy))), fwhm=4.0,
                  full_output=True, verbose=False)
    pp.run()
    assert pp.final_residuals_cube.shape == (4, 48, 48) and pp.frame_final.shape == (48, 48)
    # 4-D standard and rotation-threshold layouts
    g4 = load_golden("g6_pca_4d")
    pp = PostProc(cube=g4["cube"], angle_list=g4["angles"], ncomp=2, full_output=True, verbose=False)
    pp.run()
    assert np.abs(pp.frame_final - g4["frame"]).max() < TOL and pp.pcs.shape[0] == g4["cube"].shape[0]
    pp = PostProc(cube=g4["cube"], angle_list=g4["angles"], ncomp=2, source_xy=(25, 16), fwhm=4, delta_rot=0.3,
                  min_frames_pca=2, full_output=True, verbose=False)
    pp.run()
    assert pp.cube_residuals.shape == g4["cube"].shape
    # ADI+mSDI double pass
    gm = load_golden("g10_msdi")
    pp = PostProc(cube=gm["cube"], angle_list=gm["angles"], scale_list=gm["scale_list"], ncomp=(2, 2), adimsdi="double",
                  full_output=True, verbose=False)
    pp.run()
    assert pp.frame_final.ndim == 2 and pp.cube_residuals.ndim == 3
    # frame only: every mode returns the bare frame (pppca.py uses full_output=True, contrast curves do not)
    assert pca(algo_params=PostProc(ncomp=3, cube=cube, angle_list=ang, verbose=False)).shape == cube.shape[1:]


def test_raw_c_abi_pca_4d():
    """vipmi_pca_4d_f32 by raw ctypes (SURVEY 8(b) minimum surface): the 4-D per-channel PCA + spectral collapse of
    reference pca_fullfr.py:544-658 in one C call, against the reference golden; mask + scaling + median against the
    Python front end's per-channel loop."""
    import ctypes
    import torch
    from conftest import ROOT
    from vip_amd.psfsub import pca
    from vip_amd.var.shapes import center_mask_u8
    g = load_golden("g6_pca_4d")
    cube_np, angle_list = g["cube"], g["angles"]
    nch, n, N = cube_np.shape[0], cube_np.shape[1], cube_np.shape[2]
    lib = ctypes.CDLL(os.path.join(ROOT, "vip_amd", "libvipmi.so"))
    lib.vipmi_last_error.restype = ctypes.c_char_p
    ctx = ctypes.c_void_p()
    assert lib.vipmi_create(0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(ctx)) == 0
    cube = torch.from_numpy(cube_np).cuda()
    frame = torch.empty((N, N), dtype=torch.float32, device="cuda")
    ifs = torch.empty((nch, N, N), dtype=torch.float32, device="cuda")
    angles = np.ascontiguousarray(O.check_pa_vector(angle_list), dtype=np.float64)
    lib.vipmi_pca_4d_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 4 + [ctypes.c_int, ctypes.c_void_p,
                                                                                     ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 2
    st = lib.vipmi_pca_4d_f32(ctx, cube.data_ptr(), angles.ctypes.data, nch, n, N, 2, 0, None, 0, 1, frame.data_ptr(),
                              ifs.data_ptr())
    assert st == 0, lib.vipmi_last_error().decode()
    torch.cuda.synchronize()
    assert np.abs(frame.cpu().numpy() - g["frame"]).max() < TOL
    assert np.abs(ifs.cpu().numpy() - g["ifs"]).max() < TOL
    mask = torch.from_numpy(center_mask_u8((N, N), 3)).cuda()
    st = lib.vipmi_pca_4d_f32(ctx, cube.data_ptr(), angles.ctypes.data, nch, n, N, 2, 1, mask.data_ptr(), 1, 0,
                              frame.data_ptr(), None)                     # temp-mean, mask, mean collapse, median over channels
    assert st == 0, lib.vipmi_last_error().decode()
    torch.cuda.synchronize()
    ref = pca(cube_np, angle_list, ncomp=2, scaling="temp-mean", mask_center_px=3, collapse="mean", collapse_ifs="median",
              full_output=True, verbose=False)[0]
    assert np.abs(frame.cpu().numpy() - ref).max() < 2e-5
    st = lib.vipmi_pca_4d_f32(ctx, cube.data_ptr(), angles.ctypes.data, nch, n, N, 0, 0, None, 0, 1, frame.data_ptr(), None)
    assert st < 0 and b"PCs" in lib.vipmi_last_error()
    lib.vipmi_destroy.argtypes = [ctypes.c_void_p]
    assert lib.vipmi_destroy(ctx) == 0


@pytest.mark.parametrize("N", [255, 301, 601])
def test_pca_odd_frame_sizes_end_to_end(N):
    """VIP's convention is ODD frames centred on the star: the whole pipeline at sizes whose derotation runs through the
    power-of-two convolution passes (derotate_conv.inc), with and without a central mask (mask_val = 0 restore), against
    the oracle."""
    from vip_amd.psfsub import pca, pca_annular
    nfr = 10 if N < 500 else 6                           # (601 px: two parts per line in the convolution passes)
    cube, ang = O.synth_adi(nfr, N, seed=8)
    ang = np.linspace(-30, 250, nfr)                     # every rot90 quadrant
    out = pca(cube, ang, ncomp=3, full_output=True, verbose=False)
    ref = O.pca_fullframe(cube, ang, ncomp=3, full_output=True)
    assert np.abs(out[0] - ref[0]).max() < TOL and np.nanmax(np.abs(out[4] - ref[4])) < TOL
    fm = pca(cube, ang, ncomp=2, mask_center_px=5, collapse="mean", verbose=False)
    assert np.abs(fm - O.pca_fullframe(cube, ang, ncomp=2, mask_center_px=5, collapse="mean")).max() < TOL
    if N == 255:
        fa = pca_annular(cube, ang, ncomp=2, asize=32, fwhm=4, verbose=False)
        assert np.nanmax(np.abs(fa - O.pca_annular(cube, ang, ncomp=2, asize=32, fwhm=4))) < TOL


def test_context_cache_eviction_keeps_held_contexts_alive(monkeypatch):
    """round-2 ADVICE: evicting the least recently used context must not invalidate a handle other code still holds
    (it is trimmed -- workspaces freed, handle valid -- and re-allocates on its next call), and a context inside a call
    is never touched."""
    import torch
    from vip_amd import backend as B
    from vip_amd.psfsub import pca
    cube, ang = O.synth_adi(20, 64, seed=5)
    ct = torch.from_numpy(cube).cuda()
    ref = pca(ct, ang, ncomp=3, verbose=False).clone()
    B.release_workspaces()
    monkeypatch.setattr(B, "MAX_CONTEXTS", 2)
    held = []
    streams = [torch.cuda.Stream() for _ in range(5)]
    for st in streams:
        with torch.cuda.stream(st):
            held.append(B.get_context())
            assert torch.equal(pca(ct, ang, ncomp=3, verbose=False), ref)
    torch.cuda.synchronize()
    assert len(B.all_contexts()) <= 2
    assert all(c.handle for c in held)                     # evicted, not destroyed
    for st, c in zip(streams, held):                       # an evicted context still works (and gives the same frame)
        with torch.cuda.stream(st):
            out = B.empty((64, 64))
            c.call("vipmi_collapse_f32", B.ptr(ct), 20, 64 * 64, 1, None, 0, B.ptr(out))
    torch.cuda.synchronize()
    # a context that is inside a call cannot be trimmed
    c = held[0]
    assert c._in_call.acquire(blocking=False)
    try:
        assert c.trim() is False
    finally:
        c._in_call.release()
    assert c.trim() is True
    B.release_workspaces()
    assert torch.equal(pca(ct, ang, ncomp=3, verbose=False), ref)


def test_async_mode_is_thread_local():
    """round-2 ADVICE: a pipelined region on one thread must not switch a concurrent caller on another thread to
    deferred error checks (nor have the mode switched off under it)."""
    import threading
    import torch
    from vip_amd import backend as B
    from vip_amd.psfsub import pca
    cube, ang = O.synth_adi(20, 64, seed=6)
    ct = torch.from_numpy(cube).cuda()
    ref = pca(ct, ang, ncomp=3, verbose=False).clone()
    seen = {}
    go, done = threading.Event(), threading.Event()

    def other():
        torch.cuda.set_device(0)
        go.wait(30)
        seen["async_in_other_thread"] = B.is_async()
        with torch.cuda.stream(torch.cuda.Stream()):
            c = B.get_context()
            seen["frame_ok"] = bool(torch.equal(pca(ct, ang, ncomp=3, verbose=False), ref))
            seen["eigh_check"] = c.get_option("eigh_check") != 0      # (-1: never set = the default, checks on)
        done.set()

    th = threading.Thread(target=other)
    th.start()
    B.set_async(True)
    try:
        go.set()
        assert done.wait(120)
        assert B.is_async()
        with torch.cuda.stream(torch.cuda.Stream()):
            out = pca(ct, ang, ncomp=3, verbose=False)
            assert B.get_context().get_option("eigh_check") == 0
        torch.cuda.synchronize()
        B.check_deferred()
    finally:
        B.set_async(False)
    th.join()
    assert seen == {"async_in_other_thread": False, "frame_ok": True, "eigh_check": True}
    assert torch.equal(out, ref)


def test_mask_val_other_than_nan_or_zero_golden():
    """G25 (round 3; round-2 VERDICT missing #4): frame_rotate's ``mask_val`` as any number on the device path --
    pixels equal to it take part in the rotation and are reset afterwards, NaNs are rotated as 0 and stay finite
    (preproc/derotation.py:133-140,324-326); through cube_derotate, frame_rotate, pca(rot_options) and median_sub."""
    from vip_amd.preproc import cube_derotate, frame_rotate
    from vip_amd.psfsub import pca, median_sub
    g = load_golden("g25_mask_val")

    def close(a, b, tol):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b))
        assert np.nanmax(np.abs(a.astype(np.float64) - b), initial=0.0) < tol

    for N in (40, 128, 45):
        fr, angs = g["in_%d" % N], g["angles_%d" % N]
        for method in ("auto", "direct"):
            close(cube_derotate(fr, angs, mask_val=5.0, method=method), g["out5_%d" % N], 3e-5)
            close(cube_derotate(fr, angs, mask_val=-1.5, method=method), g["outm_%d" % N], 3e-5)
        out = cube_derotate(fr, angs, mask_val=5.0)
        assert np.array_equal(out == 5.0, g["out5_%d" % N] == 5.0)            # the reset pixels, bit-exact
        close(frame_rotate(fr[0], 33.0, mask_val=5.0), g["fr5_%d" % N], 3e-5)
    close(cube_derotate(g["in_01"], np.array([20.0, -50.0]), mask_val=0.1), g["out_01"], 3e-5)
    cube, ang = g["cube"], g["angles"]
    fo = pca(cube, ang, ncomp=3, mask_center_px=5, mask_val=2.5, full_output=True, verbose=False)
    close(fo[0], g["pca_frame"], TOL)
    close(fo[4], g["pca_resder"], TOL)
    close(pca(cube, ang, ncomp=2, mask_val=7.0, verbose=False), g["pca_nomask"], TOL)
    close(median_sub(cube, ang, mask_val=1e3, verbose=False), g["medsub"], TOL)


def test_pca_grid_scored_by_snr_on_a_4d_cube():
    """round-2 VERDICT missing #5: the S/N-scored grid on a 4-D cube without scale_list (reference
    psfsub/pca_fullfr.py:604-612,619-623,779-786): a grid per channel, the channel's frame = its best-S/N frame, the
    final frame = spectral collapse of those; full_output -> (per-channel grid cubes, frame, tables, ifs_adi_frames)."""
    from vip_amd.psfsub import pca
    chans = [_cube_with_companion(seed=9 + i, flux=3.0 - 0.5 * i) for i in range(3)]
    c4 = np.stack([c[0] for c in chans])
    ang = chans[0][1]
    sx, sy = chans[0][2]
    xy = (float(round(sx)), float(round(sy)))
    refs = [O.pca_grid_snr(c4[ch], ang, (1, 5, 1), xy, 4.0) for ch in range(3)]
    out = pca(c4, ang, ncomp=(1, 5, 1), source_xy=xy, fwhm=4.0, full_output=True, verbose=False)
    assert len(out) == 4
    cubes, frame, tables, ifs = out
    assert cubes.shape == (3, 5, 48, 48) and ifs.shape == (3, 48, 48) and len(tables) == 3
    for ch in range(3):
        assert np.nanmax(np.abs(cubes[ch] - refs[ch][0])) < TOL
        assert list(tables[ch]["PCs"]) == refs[ch][2]
        assert np.nanmax(np.abs(ifs[ch] - refs[ch][1])) < TOL
    exp = O.cube_collapse(np.stack([r[1] for r in refs]), "mean")
    assert frame.dtype == np.float64 and np.nanmax(np.abs(frame - exp)) < TOL
    only = pca(c4, ang, ncomp=(1, 5, 1), source_xy=xy, fwhm=4.0, verbose=False, collapse_ifs="median")
    assert np.nanmax(np.abs(only - O.cube_collapse(np.stack([r[1] for r in refs]), "median"))) < TOL
    # the per-channel grid cubes are those of the unscored grid, which the reference-generated fixtures pin
    # (a list ncomp whose length differs from the number of channels is a grid for every channel, :548-551)
    fo = pca(c4, ang, ncomp=[1, 2, 3, 4, 5], full_output=True, verbose=False)
    assert np.nanmax(np.abs(fo[2] - cubes)) < 1e-6


G26_CASES = (("rsdi", dict(ncomp=(2, 3), ref=True)),
             ("rsdi_tm", dict(ncomp=(1, 2), ref=True, scaling="temp-mean", collapse="mean", mask_center_px=3)),
             ("rsdi_noadi", dict(ncomp=(2, None), ref=True)),
             ("thr", dict(ncomp=(2, 3), source_xy=(24.0, 20.0), delta_rot=0.5, fwhm=4.0, min_frames_pca=3)),
             ("thr_ref", dict(ncomp=(2, 4), source_xy=(22.0, 9.0), delta_rot=1.0, fwhm=4.0, min_frames_pca=3, ref=True,
                              max_frames_pca=5)),
             ("thr_aref", dict(ncomp=(None, 3), source_xy=(8.0, 18.0), delta_rot=0.8, fwhm=4.0, min_frames_pca=3, ref=True,
                               ref_strategy="ARSDI")))


@pytest.mark.parametrize("tag,kw", G26_CASES)
def test_msdi_double_with_reference_cube_and_rotation_threshold_golden(tag, kw):
    """round-2 VERDICT missing #5: ADI+mSDI double pass with ``cube_ref`` (RSDI library of the second stage) and / or a
    rotation threshold at ``source_xy`` (reference psfsub/pca_fullfr.py:1279-1283,1388-1459), against the reference's own
    outputs (G26)."""
    from vip_amd.psfsub import pca
    g = load_golden("g26_msdi_double_ref_thr")
    kw = dict(kw)
    if kw.pop("ref", False):
        kw["cube_ref"] = g["cube_ref"]
    fo = pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="double", full_output=True, verbose=False, **kw)
    assert len(fo) == 3
    for nm, a in zip(("frame", "chan", "chan_der"), fo):
        exp = g["%s_%s" % (tag, nm)]
        assert a.shape == exp.shape, (tag, nm, a.shape, exp.shape)
        assert np.array_equal(np.isnan(a), np.isnan(exp)), (tag, nm)
        assert np.nanmax(np.abs(a - exp)) < TOL, (tag, nm, np.nanmax(np.abs(a - exp)))
    only = pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="double", verbose=False, **kw)
    assert np.array_equal(only, fo[0], equal_nan=True)


def test_pca_3d_rotation_threshold_with_a_reference_cube():
    """source_xy + cube_ref in the 3-D branch (pca_fullfr.py:911-965 with _project_subtract's indices + cube_ref mode,
    :1693-1694): every frame's library = its rotation-compliant frames + all reference frames; against the oracle, whose
    per-frame mode G26 pins through the double pass."""
    from vip_amd.psfsub import pca
    cube, ang = O.synth_adi(16, 40, seed=260)
    cref = O.synth_adi(7, 40, seed=261)[0]
    for kw in (dict(ncomp=3), dict(ncomp=4, scaling="temp-standard", max_frames_pca=6)):
        exp = O.pca_pa_rejection(cube, ang, source_xy=(28.0, 24.0), fwhm=4.0, delta_rot=0.7, min_frames_pca=3,
                                 cube_ref=cref, full_output=True, **kw)
        out = pca(cube, ang, source_xy=(28.0, 24.0), fwhm=4.0, delta_rot=0.7, min_frames_pca=3, cube_ref=cref,
                  full_output=True, verbose=False, **kw)
        # (frame, recon_cube, residuals_cube, residuals_cube_)
        for a, b in zip(out, exp):
            assert a.shape == b.shape and np.nanmax(np.abs(a - b)) < TOL


G27_CASES = (("plain", dict(ncomp=(2, 3))), ("rsdi", dict(ncomp=(2, 3), ref=True, scaling="temp-mean")),
             ("thr", dict(ncomp=(1, 3), source_xy=(23.0, 16.0), delta_rot=0.6, fwhm=4.0, min_frames_pca=3)))


@pytest.mark.parametrize("tag,kw", G27_CASES)
def test_msdi_double_with_cube_sig_golden(tag, kw):
    """``cube_sig`` in the ADI+mSDI double pass (handed to the second stage's ``_project_subtract``, reference
    psfsub/pca_fullfr.py:1395,1409,1447) against the reference's own outputs (G27)."""
    from vip_amd.psfsub import pca
    g = load_golden("g27_msdi_double_sig")
    kw = dict(kw)
    if kw.pop("ref", False):
        kw["cube_ref"] = g["cube_ref"]
    fo = pca(g["cube"], g["angles"], scale_list=g["scale_list"], adimsdi="double", full_output=True, verbose=False,
             cube_sig=g["cube_sig"], **kw)
    for nm, a in zip(("frame", "chan", "chan_der"), fo):
        exp = g["%s_%s" % (tag, nm)]
        assert a.shape == exp.shape and np.array_equal(np.isnan(a), np.isnan(exp)), (tag, nm)
        assert np.nanmax(np.abs(a - exp)) < TOL, (tag, nm, np.nanmax(np.abs(a - exp)))


@pytest.mark.parametrize("tag,kw", [("k4", dict(ncomp=4)), ("k4_tm", dict(ncomp=4, scaling="temp-mean")),
                                    ("k9_mask", dict(ncomp=9, mask_center_px=6))])
def test_float64_cube_of_detector_counts_vs_reference(tag, kw):
    """A float64 cube with values ~7e3 (g28: the reference's OWN float64 run, oracle/gen_golden_r5.py).  The reference keeps the
    caller's dtype through svd_wrapper (psfsub/pca_fullfr.py:1552-1737).  The float64 route of the device path
    (vipmi_pca_fullframe_f64, csrc/pca_f64.hip) carries the per-pixel temporal mean in float64 and runs the float32 kernels on what
    is left: the BASELINE gate 1e-4 holds on this cube although its samples are 700 times the benchmark's (measured 2.2e-5 .. 3.5e-5;
    the float32 route -- the cube rounded on upload -- ends 1.9e-3 away, the reference itself 1.6e-3 .. 4.7e-2 when it is handed the
    cube as float32)."""
    import torch
    from vip_amd.psfsub import pca
    g = load_golden("g28_f64_counts")
    cube, ang = g["cube"], g["angles"]
    assert cube.dtype == np.float64
    fr = pca(cube, ang, verbose=False, **kw)
    assert fr.dtype == np.float64                         # numpy in -> numpy out in the caller's dtype
    dev = np.nanmax(np.abs(fr - g["frame64_" + tag]))
    ref32 = np.nanmax(np.abs(g["frame_ref_f32_" + tag] - g["frame64_" + tag]))
    print("g28 %s: float64 route vs reference(f64) %.3e; reference(f32 cube) vs reference(f64) %.3e" % (tag, dev, ref32))
    assert dev < TOL and dev < 0.2 * ref32
    # a float64 cuda tensor takes the same route: bit-identical frame, returned on the device
    frt = pca(torch.from_numpy(cube).cuda(), ang, verbose=False, **kw)
    assert frt.is_cuda and np.array_equal(np.nan_to_num(frt.cpu().numpy().astype(np.float64), nan=3.5), np.nan_to_num(fr, nan=3.5))
    # full_output through the same route: frame, pcs, recon, residuals, derotated residuals against the float64 oracle
    fo = pca(cube, ang, verbose=False, full_output=True, **kw)
    ro = O.pca_fullframe(cube, ang, full_output=True, **kw)
    assert np.array_equal(np.nan_to_num(fo[0], nan=3.5), np.nan_to_num(fr, nan=3.5))
    assert np.nanmax(np.abs(fo[3] - g["res64_" + tag])) < 2 * TOL                       # the reference's own residual cube
    for nm, a, b in zip(("frame", "pcs", "recon", "res", "resder"), fo, ro):
        assert a.dtype == np.float64 and a.shape == b.shape, nm
        if nm == "pcs":
            a = sign_align(a, b)
        tol = {"pcs": 1e-6, "recon": 2e-3}.get(nm, 2 * TOL)          # (recon holds the counts themselves: float32 of 7e3)
        assert np.nanmax(np.abs(a - b)) < tol, (nm, np.nanmax(np.abs(a - b)))
    # the float32 route (what every other call shape still takes for float64 input): bounded by 2^-21 of the largest sample
    f32r = pca(cube, ang, verbose=False, cube_sig=np.zeros_like(cube), **kw) if "mask_center_px" not in kw else None
    if f32r is not None:
        assert np.nanmax(np.abs(f32r - g["frame64_" + tag])) <= 2.0 ** -21 * np.abs(cube).max()


@pytest.mark.parametrize("tag,kw", [("smean", dict(ncomp=4, scaling="spat-mean")), ("sstd", dict(ncomp=5, scaling="spat-standard")),
                                    ("sstd_mask", dict(ncomp=3, scaling="spat-standard", mask_center_px=6))])
def test_float64_cube_spatial_scalings_vs_reference(tag, kw):
    """matrix_scaling(axis=1) (var/shapes.py:740-781) of a float64 cube of detector counts on the float64 route: the scaled matrix
    is D + u mu'^T with D formed in float64 and the offset mu' carried beside it (csrc/pca_f64.hip).  g30 = the reference's OWN
    float64 runs on g28's cube (oracle/gen_golden_r6.py); handed the cube as float32 the reference itself ends 3.1e-4 away for
    'spat-mean' (above the BASELINE gate), and so did the device path before round 6."""
    from vip_amd.psfsub import pca
    g28, g = load_golden("g28_f64_counts"), load_golden("g30_f64_spat")
    cube, ang = g28["cube"], g28["angles"]
    fr = pca(cube, ang, verbose=False, **kw)
    assert fr.dtype == np.float64
    exp = g["frame64_" + tag]
    dev = np.nanmax(np.abs(fr - exp))
    ref32 = np.nanmax(np.abs(g["frame_ref_f32_" + tag] - exp))
    print("g30 %s: float64 route vs reference(f64) %.3e; reference(f32 cube) vs reference(f64) %.3e" % (tag, dev, ref32))
    assert np.array_equal(np.isnan(fr), np.isnan(exp))
    assert dev < TOL * max(1.0, np.nanmax(np.abs(exp)) / 10.0)
    if tag == "smean":
        assert dev < 0.2 * ref32
    fo = pca(cube, ang, verbose=False, full_output=True, **kw)
    ro = O.pca_fullframe(cube, ang, full_output=True, **kw)
    if tag == "smean":
        assert np.nanmax(np.abs(fo[3] - g["res64_smean"])) < 2 * TOL                   # the reference's own residual cube
    for nm, a, b in zip(("frame", "pcs", "recon", "res", "resder"), fo, ro):
        assert a.dtype == np.float64 and a.shape == b.shape, nm
        if nm == "pcs":
            a = sign_align(a, b)
        tol = {"pcs": 1e-6, "recon": 2e-3}.get(nm, 2 * TOL)          # (recon holds the scaled counts: float32 of up to 7e3)
        assert np.nanmax(np.abs(a - b)) < tol, (nm, np.nanmax(np.abs(a - b)))


def test_float64_route_scalings_collapses_and_fallbacks():
    """vipmi_pca_fullframe_f64 beyond the goldens: 'temp-standard', the spatial scalings with a mask / a long basis, every collapse it serves, ncomp > n clamped; the shapes it does
    not serve (cube_ref, a tuple of ncomp) fall back on the float32 route -- all against the float64 oracle."""
    from vip_amd.psfsub import pca
    g = load_golden("g28_f64_counts")
    cube, ang = g["cube"][:, 8:56, 8:56].copy(), g["angles"]
    for kw in (dict(ncomp=5, scaling="temp-standard"), dict(ncomp=3, collapse="mean"), dict(ncomp=3, collapse="max"),
               dict(ncomp=6, collapse="sum"), dict(ncomp=100), dict(ncomp=2, scaling="temp-mean", mask_center_px=4),
               dict(ncomp=3, scaling="spat-mean", mask_center_px=5, collapse="mean"), dict(ncomp=30, scaling="spat-standard")):
        ref = O.pca_fullframe(cube, ang, **{k_: min(v, cube.shape[0]) if k_ == "ncomp" else v for k_, v in kw.items()})
        fr = pca(cube, ang, verbose=False, **kw)
        scale = max(1.0, np.nanmax(np.abs(ref)))
        assert np.nanmax(np.abs(fr - ref)) < TOL * max(1.0, scale / 10.0), kw
    for kw in (dict(ncomp=3, cube_ref=cube[:9].copy()), dict(ncomp=(1, 3))):
        out = pca(cube, ang, verbose=False, **kw)          # float32 route: runs, same shapes as ever
        assert out is not None


def test_float64_route_4d_cube():
    """A 4-D float64 cube of detector counts without scale_list: every channel through the float64 route, then collapse_ifs --
    against the float64 oracle at the BASELINE gate; the float32 route (a float32 copy of the cube) is 50 times further away."""
    from vip_amd.psfsub import pca
    g = load_golden("g28_f64_counts")
    c3, ang = g["cube"][:, 4:60, 4:60], g["angles"]
    cube4 = np.stack([c3, 0.5 * c3[::-1] + 100.0, c3 * 1.25 - 50.0])
    for kw in (dict(ncomp=4), dict(ncomp=[3, 5, 4], collapse_ifs="median"), dict(ncomp=5, scaling="temp-mean", mask_center_px=4)):
        ks = kw["ncomp"] if isinstance(kw["ncomp"], list) else [kw["ncomp"]] * 3
        okw = {a: b for a, b in kw.items() if a not in ("ncomp", "collapse_ifs")}
        per = np.stack([O.pca_fullframe(cube4[ch], ang, ncomp=ks[ch], **okw) for ch in range(3)])
        ref = {"median": np.nanmedian, "mean": np.nanmean}[kw.get("collapse_ifs", "mean")](per, axis=0)
        fr = pca(cube4, ang, verbose=False, **kw)
        assert fr.dtype == np.float64 and fr.shape == ref.shape
        d64 = np.nanmax(np.abs(fr - ref))
        d32 = np.nanmax(np.abs(pca(cube4.astype(np.float32), ang, verbose=False, **kw) - ref))
        assert d64 < TOL and d32 > 10 * d64, (kw, d64, d32)


@pytest.mark.parametrize("kw", [dict(ncomp=3), dict(ncomp=(2, 3, 4, 3), delta_rot=(0.2, 0.8)), dict(ncomp=3, scaling="temp-mean"),
                                dict(ncomp=2, scaling="temp-standard"), dict(ncomp=3, n_segments=2, radius_int=4),
                                dict(ncomp=3, scaling="spat-mean"), dict(ncomp=(2, 3, 2, 4), scaling="spat-standard", n_segments=2)])
@pytest.mark.parametrize("fused", ["0", "1"])
def test_float64_route_annular(kw, fused, monkeypatch):
    """pca_annular on a float64 cube of detector counts: every segment matrix centred in float64 (vipmi_center_f64), the Gram
    matrix corrected by the offset terms, residuals = (I - C) D + rho mu^T -- against the float64 oracle at the BASELINE gate;
    the float32 route (a float32 copy of the cube) ends at least ten times further away.  fused = 1: the same through the fronts of
    all segments at once (vipmi_annular_gram_all_f64: gather + centring in one pass, one ragged Gram product, the offset terms of every
    segment, the rank-one term added as the residual product scatters; forced: the cube is below the route's size threshold).
    The spatial scalings (round 6: per segment, vipmi_spat_center_f64 -> D + u mu^T, rho = (I - C) u) keep the per-segment launches
    either way."""
    from vip_amd.psfsub import pca_annular
    monkeypatch.setenv("VIPMI_ANNULAR_FUSED", fused)
    g = load_golden("g28_f64_counts")
    cube, ang = g["cube"], np.linspace(0, 120, g["cube"].shape[0])
    base = dict(asize=8, fwhm=4, delta_rot=(0.3, 1), verbose=False)
    base.update(kw)
    okw = {a: b for a, b in base.items() if a != "verbose"}
    ref = O.pca_annular(cube, ang, **okw)
    fr = pca_annular(cube, ang, **base)
    assert fr.dtype == np.float64
    d64 = np.nanmax(np.abs(fr - ref))
    d32 = np.nanmax(np.abs(pca_annular(cube.astype(np.float32), ang, **base) - ref))
    print("annular float64 route %s: %.3e (float32 route %.3e)" % (kw, d64, d32))
    scale = max(10.0, np.nanmax(np.abs(ref)))
    assert d64 < TOL * scale / 10.0
    if kw.get("scaling") not in ("temp-standard", "spat-standard"):
        assert d32 > 5 * d64


def test_float64_route_median_sub():
    """median_sub on a float64 cube of detector counts: the per-pixel temporal mean taken off in float64 first (the subtraction of
    medians is invariant under it) -- full-frame and annular mode against the float64 oracle."""
    from vip_amd.psfsub import median_sub
    g = load_golden("g28_f64_counts")
    cube, ang = g["cube"], np.linspace(0, 120, g["cube"].shape[0])
    for kw in (dict(), dict(mode="annular", asize=8, delta_rot=0.5, nframes=4, fwhm=4)):
        if kw:
            ref = O.median_sub_annular(cube, ang, **{k: v for k, v in kw.items() if k != "mode"})
        else:
            ref = O.median_sub_fullfr(cube, ang)
        fr = median_sub(cube, ang, verbose=False, **kw)
        assert fr.dtype == np.float64
        d64 = np.nanmax(np.abs(fr - ref))
        d32 = np.nanmax(np.abs(median_sub(cube.astype(np.float32), ang, verbose=False, **kw) - ref))
        print("median_sub float64 %s: %.3e (float32 cube %.3e)" % (kw, d64, d32))
        assert d64 < TOL and d32 > 3 * d64


def test_annular_library_window_skipping_is_bit_identical():
    """The (I - C) A product of annular PCA skips the frames outside every row group's library window (`ann_range`), and the
    libraries' sub-Gram matrices are gathered inside the eigensolver (`ann_gather`): both must leave the residual cube bit-identical
    to the dense product on materialised matrices (only exact zeros are skipped; the same values reach the same kernels)."""
    import torch
    from vip_amd import backend as B
    from vip_amd.psfsub import pca_annular
    cube, ang = O.synth_adi(130, 96, seed=77)
    ct = torch.from_numpy(cube).cuda()
    ang = np.linspace(0, 140, 130)
    kw = dict(asize=12, ncomp=4, fwhm=4, delta_rot=(0.3, 1), full_output=True, verbose=False, max_frames_lib=60)
    outs = {}
    for opts in ((1, 1), (0, 1), (1, 0), (0, 0)):
        for c in [B.get_context()] + list(B.all_contexts()):
            c.set_option("ann_range", opts[0])
            c.set_option("ann_gather", opts[1])
        os.environ["VIPMI_OPTS"] = "ann_range=%d,ann_gather=%d" % opts          # (contexts of the side streams created later)
        try:
            outs[opts] = [t.clone() for t in pca_annular(ct, ang, **kw)]
        finally:
            os.environ.pop("VIPMI_OPTS", None)
    for c in B.all_contexts():
        c.set_option("ann_range", 1)
        c.set_option("ann_gather", 1)
    ref = outs[(0, 0)]
    for opts, got in outs.items():
        for a, b in zip(got, ref):
            assert torch.equal(torch.nan_to_num(a, nan=7.5), torch.nan_to_num(b, nan=7.5)), opts
    assert np.abs(ref[2].cpu().numpy() - O.pca_annular(cube, ang, asize=12, ncomp=4, fwhm=4, delta_rot=(0.3, 1), max_frames_lib=60)).max() < TOL


def test_numpy_cube_through_the_host_input_entry_is_bit_identical():
    """pca(float32 numpy cube) takes vipmi_pca_fullframe_hostin_f32 when the call is the plain ADI one and the cube is big enough for
    the int8 Gram path: the library uploads the cube in blocks of 64 frames and advances the Gram matrix behind every block.  Same
    partial sums in the same order: the frame and every full_output array equal the upload-then-call route bit for bit -- frame
    counts that are no multiple of 64, a count just above a block boundary and a masked call (every block masked as it arrives)
    included; smaller cubes and scaled calls keep the old route."""
    from vip_amd.psfsub import pca
    for n, N, k, mpx in ((320, 512, 6, None), (257, 384, 5, None), (300, 512, 4, 9)):
        cube, ang = O.synth_adi(n, N, seed=n)
        res = {}
        try:
            for h in ("0", "1"):
                os.environ["VIPMI_HOSTIN"] = h
                res[h] = pca(cube, ang, ncomp=k, mask_center_px=mpx, verbose=False, check_memory=False, full_output=True)
        finally:
            os.environ.pop("VIPMI_HOSTIN", None)
        for a, b in zip(res["0"], res["1"]):
            assert a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)
    # (not eligible: scaling changes the matrix the Gram is taken of -> the ordinary route, same result as a tensor input)
    import torch
    cube, ang = O.synth_adi(280, 512, seed=5)
    a = pca(cube, ang, ncomp=4, scaling="temp-mean", verbose=False, check_memory=False)
    b = pca(torch.from_numpy(cube).cuda(), ang, ncomp=4, scaling="temp-mean", verbose=False, check_memory=False).cpu().numpy()
    assert np.array_equal(a, b, equal_nan=True)


def test_numpy_4d_cube_uploaded_in_channel_groups_is_bit_identical():
    """pca(4-D float32 numpy cube, int ncomp, no scale_list): the channels go up in groups on a copy stream (an uploader thread)
    while the batched per-channel path works on the group before; the Gram launch of every group is pinned to the split-K
    slicing of the whole batch, everything else is per problem / per frame: the frame equals upload-then-call bit for bit,
    with and without mask_center_px."""
    from vip_amd.psfsub import pca
    rng = np.random.default_rng(4)
    for nch, n, N, k, mpx in ((13, 140, 256, 4, None), (9, 130, 320, 6, 5)):
        cube = rng.standard_normal((nch, n, N, N), dtype=np.float32) + rng.standard_normal((nch, 1, N, N), dtype=np.float32) * 3
        ang = np.linspace(0, 90, n)
        res = {}
        try:
            for h in ("0", "1"):
                os.environ["VIPMI_HOSTIN"] = h
                res[h] = pca(cube, ang, ncomp=k, mask_center_px=mpx, verbose=False, check_memory=False)
        finally:
            os.environ.pop("VIPMI_HOSTIN", None)
        assert res["0"].dtype == res["1"].dtype and np.array_equal(res["0"], res["1"], equal_nan=True)
