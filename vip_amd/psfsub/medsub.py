"""Median-ADI / median-RDI: drop-in for ``vip_hci.psfsub.median_sub`` (reference psfsub/medsub.py:60-88
MEDIAN_SUB_Params, :91-519 median_sub; full-frame branch :279-319, annular branch :316-371 with
_median_subt_ann_adi / _median_subt_ann_rdi :602-676, :376-387, :516-519).  Composed from the device
kernels of the PCA path: NaN-aware median over the frames, subtraction (the project/subtract kernel with a single
all-ones coefficient), annulus gather / scatter, a small per-frame subset median, FFT derotation, collapse.

Not accelerated (NotImplementedError): 4-D (SDI) cubes, flux-scaled reference subtraction (``collapse_ref`` starting
with ``sc``), ``mode='annular'`` with ``nframes=None`` or more than 32 frames per optimised reference.
"""
from dataclasses import dataclass
from enum import Enum
from typing import List, Tuple, Union

import numpy as np

from .. import backend as B
from ..config.paramenum import ALGO_KEY, Collapse, Imlib, Interpolation
from ..config.utils_param import separate_kwargs_dict
from ..preproc.derotation import _define_annuli, _find_indices_adi_nframes_all
from ..preproc.parangles import check_pa_vector
from ..var.shapes import center_mask_u8, get_annulus_segments


@dataclass
class MEDIAN_SUB_Params:
    """Parameters of ``median_sub`` (field order == positional order of the reference)."""

    cube: np.ndarray = None
    angle_list: np.ndarray = None
    scale_list: np.ndarray = None
    flux_sc_list: np.ndarray = None
    fwhm: float = 4
    radius_int: int = 0
    asize: int = 4
    delta_rot: int = 1
    delta_sep: Union[float, Tuple[float]] = (0.1, 1)
    mode: str = "fullfr"
    nframes: int = 4
    sdi_only: bool = False
    imlib: Enum = Imlib.VIPFFT
    interpolation: Enum = Interpolation.LANCZOS4
    collapse: Enum = Collapse.MEDIAN
    cube_ref: np.ndarray = None
    collapse_ref: str = "median"
    nproc: int = 1
    full_output: bool = False
    verbose: bool = True


def _s(x):
    return str(getattr(x, "value", x)) if x is not None else None


def _annular_pass(cube_t, sub_t, angle_list, algo_params, rdi):
    """mode='annular' (reference medsub.py:316-371).  ``sub_t`` = cube minus the global model (median of the cube for
    ADI, collapsed reference for RDI).  ADI: in every annulus each frame additionally loses the median of the
    ``nframes`` frames closest in time beyond the annulus' PA threshold (:602-641); RDI: the annuli keep ``sub_t``
    (:644-676).  Pixels outside the annuli are zero."""
    torch = B._torch()
    n, y, x = sub_t.shape
    P = y * x
    dev = sub_t.device.index
    ctx = B.get_context(dev)
    radius_int, asize = algo_params.radius_int, algo_params.asize
    n_annuli = int((y / 2 - radius_int) / asize)
    if algo_params.verbose:
        print("N annuli = {}, FWHM = {}".format(n_annuli, algo_params.fwhm))
    nframes = algo_params.nframes
    if not rdi and nframes is not None and nframes % 2 != 0:
        raise TypeError("`nframes` argument must be even value")
    cube_out = torch.zeros_like(sub_t)
    for ann in range(n_annuli):
        if rdi:
            inner_radius = radius_int + ann * asize            # (no last-annulus rule in the RDI branch, :650)
            pa_thr = 0
        else:
            pa_thr, inner_radius, _ = _define_annuli(angle_list, ann, n_annuli, algo_params.fwhm, radius_int, asize,
                                                     algo_params.delta_rot, 1, False)
        yy, xx = get_annulus_segments((y, x), inner_radius, asize, 1)[0]
        pix = torch.from_numpy((yy.astype(np.int64) * x + xx).astype(np.int32)).to(sub_t.device)
        npx = int(pix.numel())
        if npx == 0:
            continue
        A = B.empty((n, npx), device=dev)
        ctx.call("vipmi_gather_f32", B.ptr(sub_t), n, P, B.ptr(pix), npx, B.ptr(A))
        if not rdi:
            if pa_thr != 0:
                if nframes is None:
                    raise NotImplementedError("median_sub(mode='annular', nframes=None) is not accelerated")
                libs = _find_indices_adi_nframes_all(angle_list, pa_thr, nframes)
            else:
                libs = [np.arange(n, dtype=np.int32) for _ in range(n)]
            wmax = max(1, max(len(li) for li in libs))
            if wmax > 32:
                raise NotImplementedError("median_sub(mode='annular'): libraries of more than 32 frames "
                                          "(nframes={}) are not accelerated".format(nframes))
            idx = np.zeros((n, wmax), dtype=np.int32)
            ln = np.zeros(n, dtype=np.int32)
            for fr, li in enumerate(libs):
                idx[fr, :len(li)] = li
                ln[fr] = len(li)
            R = B.empty((n, npx), device=dev)
            idx_t, ln_t = torch.from_numpy(idx).to(sub_t.device), torch.from_numpy(ln).to(sub_t.device)   # (kept alive)
            ctx.call("vipmi_subset_median_sub_f32", B.ptr(A), n, npx, B.ptr(idx_t), B.ptr(ln_t), wmax, B.ptr(R))
            A = R
        ctx.call("vipmi_scatter_f32", B.ptr(A), n, P, B.ptr(pix), npx, B.ptr(cube_out))
    return cube_out


def median_sub(*all_args: List, **all_kwargs: dict):
    """Median PSF subtraction of a 3-D ADI cube on the MI355X.  Returns ``frame`` or
    ``(cube_out, cube_der, frame)``."""
    class_params, rot_options = separate_kwargs_dict(initial_kwargs=all_kwargs, parent_class=MEDIAN_SUB_Params)
    algo_params = None
    if ALGO_KEY in rot_options.keys():
        algo_params = rot_options[ALGO_KEY]
        del rot_options[ALGO_KEY]
    if algo_params is None:
        algo_params = MEDIAN_SUB_Params(*all_args, **class_params)
    # by default, interpolate masked area before derotation if a mask is used (medsub.py:226-229)
    if algo_params.radius_int and len(rot_options) == 0:
        rot_options["mask_val"] = 0
        rot_options["ker"] = 1
        rot_options["interp_zeros"] = True
    cube = algo_params.cube
    if not (isinstance(cube, np.ndarray) or B.is_device_tensor(cube)) or cube.ndim not in (3, 4):
        raise TypeError("Input array is not a 3d or 4d array")
    if cube.ndim == 4:
        raise NotImplementedError("4-D (SDI) median subtraction is not accelerated")
    rot_mode = B.rotation_mode(algo_params.imlib, algo_params.interpolation, rot_options.get("border_mode", "constant"), rot_options.get("mask_val"))    # 'vip-fft' or 'opencv' (medsub.py:376-387)
    if algo_params.mode not in ("fullfr", "annular"):
        raise RuntimeError("Mode not recognized")
    annular = algo_params.mode == "annular"
    torch = B._torch()
    angle_list = check_pa_vector(np.asarray(algo_params.angle_list, dtype=np.float64))
    dev_in = B.is_device_tensor(cube)
    out_dtype = None if dev_in else (cube.dtype if cube.dtype.kind == "f" else np.float64)
    is64 = (cube.dtype == torch.float64) if dev_in else (cube.dtype == np.float64)
    if is64 and algo_params.cube_ref is None:
        # a float64 cube (the reference subtracts in the caller's dtype): cube - median(cube) does not change when a per-pixel
        # constant is taken off every frame first, so the temporal mean is removed in float64 (vipmi_center_f64) and the float32
        # kernels subtract medians of what is left -- counts of 7e3 rounded to float32 first would cost 4e-4 per sample
        c64 = (cube if dev_in else torch.from_numpy(np.ascontiguousarray(cube))).to(torch.device("cuda", torch.cuda.current_device()))
        c64 = c64.contiguous()
        n_, y_, x_ = c64.shape
        t = B.empty((n_, y_, x_), device=c64.device.index)
        mu = torch.empty((y_ * x_,), dtype=torch.float64, device=c64.device)
        B.get_context(c64.device.index).call("vipmi_center_f64", B.ptr(c64), n_, y_ * x_, 1, B.ptr(t), B.ptr(mu), None)
        del c64
    else:
        t = B.to_device_f32(cube)
    n, y, x = t.shape
    P = y * x
    if algo_params.cube_ref is not None:
        ref = B.to_device_f32(algo_params.cube_ref)
        if ref.shape[-1] != x or ref.shape[-2] != y:
            raise TypeError("Reference cube shape should have same xy dimensions as science cube")
        cref = algo_params.collapse_ref
        if "sc" in cref:
            raise NotImplementedError("flux-scaled reference subtraction (collapse_ref='sc_...') is not accelerated")
        if "median" in cref:
            model = B.collapse(ref, "median")
        elif "mean" in cref:
            model = B.collapse(ref, "mean")
        else:
            raise NotImplementedError("collapse_ref must contain 'median' or 'mean' on the device path")
    if n != angle_list.shape[0]:
        raise TypeError("Input vector or parallactic angles has wrong length")
    if algo_params.cube_ref is None:
        model = B.collapse(t, "median")          # np.median of the cube (medsub.py:279-280)
    # cube_out = cube - model : the subtract kernel with one "component" and unit coefficients
    ctx = B.get_context(t.device.index)
    ones = torch.ones((n, 1), dtype=torch.float32, device=t.device)
    cube_out = B.empty((n, P), device=t.device.index)
    model_row = model.reshape(1, P).contiguous()          # (named: a temporary must outlive the enqueued kernel's launch)
    # (without the projection's zero guard: for an odd frame count the median IS one of the samples and the reference's
    #  `cube - median` is exactly 0 there; with radius_int > 0 its mask_val = 0 rotation resets exactly those pixels, medsub.py:279-285)
    ctx.set_option("sub_guard", 0)
    try:
        ctx.call("vipmi_subtract_gemm_f32", B.ptr(t.reshape(n, P)), B.ptr(ones), B.ptr(model_row),
                 n, 1, P, B.ptr(cube_out), None)
    finally:
        ctx.set_option("sub_guard", 1)
    cube_out = cube_out.reshape(n, y, x)
    if annular:
        cube_out = _annular_pass(t, cube_out, angle_list, algo_params, rdi=algo_params.cube_ref is not None)
    if algo_params.verbose:
        print("Optimized median psf reference subtracted" if annular else "Median psf reference subtracted")
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    with rot_mode:
        cube_der = B.derotate(cube_out, angle_list, mask_nan=mv_nan, mask_zero=not mv_nan)
    if algo_params.radius_int:
        mask = B.to_device_f32(center_mask_u8((y, x), algo_params.radius_int).astype(np.float32)).to(torch.uint8)
        cube_out = B.apply_mask(cube_out.reshape(n, -1), mask.reshape(-1), 0.0).reshape(n, y, x)
        cube_der = B.apply_mask(cube_der.reshape(n, -1), mask.reshape(-1), 0.0).reshape(n, y, x)
    collapse = _s(algo_params.collapse)
    if collapse not in B.COLLAPSE_MODES or collapse == "stim":
        raise TypeError("mode not recognized")
    frame = B.collapse(cube_der, collapse)
    if algo_params.verbose:
        print("Done derotating and combining")

    def host(v):
        return v if dev_in else B.to_host(v, out_dtype)

    if algo_params.full_output:
        return host(cube_out), host(cube_der), host(frame)
    return host(frame)
