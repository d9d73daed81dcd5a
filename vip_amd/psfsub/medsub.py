"""Median-ADI / median-RDI, full-frame mode: drop-in for ``vip_hci.psfsub.median_sub`` (reference
psfsub/medsub.py:60-88 MEDIAN_SUB_Params, :91-519 median_sub; full-frame branch :279-319, :376-387, :516-519;
SURVEY 8(f) #3).  Composed from the device kernels of the PCA path: NaN-aware median over the frames, subtraction
(the project/subtract kernel with a single all-ones coefficient), FFT derotation, collapse.

Not accelerated (NotImplementedError): ``mode='annular'``, 4-D (SDI) cubes, flux-scaled reference subtraction
(``collapse_ref`` starting with ``sc``).
"""
from dataclasses import dataclass
from enum import Enum
from typing import List, Tuple, Union

import numpy as np

from .. import backend as B
from ..config.paramenum import ALGO_KEY, Collapse, Imlib, Interpolation
from ..config.utils_param import separate_kwargs_dict
from ..preproc.parangles import check_pa_vector
from ..var.shapes import center_mask_u8


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
    if _s(algo_params.imlib) != "vip-fft":
        raise NotImplementedError("vip_amd implements imlib='vip-fft' only")
    if algo_params.mode == "annular":
        raise NotImplementedError("median_sub(mode='annular') is not accelerated")
    if algo_params.mode != "fullfr":
        raise RuntimeError("Mode not recognized")
    torch = B._torch()
    angle_list = check_pa_vector(np.asarray(algo_params.angle_list, dtype=np.float64))
    dev_in = B.is_device_tensor(cube)
    out_dtype = None if dev_in else (cube.dtype if cube.dtype.kind == "f" else np.float64)
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
    ctx.call("vipmi_subtract_gemm_f32", B.ptr(t.reshape(n, P)), B.ptr(ones), B.ptr(model.reshape(1, P).contiguous()),
             n, 1, P, B.ptr(cube_out), None)
    cube_out = cube_out.reshape(n, y, x)
    if algo_params.verbose:
        print("Median psf reference subtracted")
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    if not mv_nan and mask_val != 0:
        raise NotImplementedError("mask_val must be np.nan or 0 on the device path")
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
        return v if dev_in else v.cpu().numpy().astype(out_dtype, copy=False)

    if algo_params.full_output:
        return host(cube_out), host(cube_der), host(frame)
    return host(frame)
