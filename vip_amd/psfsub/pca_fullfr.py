"""Full-frame PCA PSF subtraction: drop-in for ``vip_hci.psfsub.pca`` (reference
psfsub/pca_fullfr.py:93-134 PCA_Params, :137-798 pca, :801-1035 _adi_rdi_pca, :1552-1737
_project_subtract) for the ADI / RDI (3-D) and per-channel (4-D, no ``scale_list``) branches.

Same positional order (= ``PCA_Params`` field order), same kwargs (unknown kwargs become
``rot_options``), same return tuples and shapes.  All array work runs on the MI355X through
libvipmi.so; numpy in -> numpy out, cuda tensor in -> cuda tensors out (no host copies).

Not accelerated (raise NotImplementedError, SURVEY.md 8(f) "next"): ``scale_list`` (mSDI), tuple/list
``ncomp`` (pca_grid), ``source_xy``, ``batch`` (incremental PCA), ``left_eigv``, ``cube_sig``,
``mask_rdi``, ``smooth``, ``imlib != 'vip-fft'``.
"""
from dataclasses import dataclass
from enum import Enum
from typing import List, Tuple, Union

import numpy as np

from .. import backend as B
from ..config.paramenum import ALGO_KEY, Adimsdi, Collapse, Imlib, Interpolation, SvdMode
from ..config.utils_param import separate_kwargs_dict, setup_parameters
from ..preproc.parangles import check_pa_vector
from ..var.shapes import center_mask_u8
from .svd import SVD_MODES, SVDecomposer


@dataclass
class PCA_Params:
    """Parameters of ``pca`` (field order == positional order of the reference)."""

    cube: np.ndarray = None
    angle_list: np.ndarray = None
    cube_ref: np.ndarray = None
    scale_list: np.ndarray = None
    ncomp: Union[Tuple, List, float, int] = 1
    svd_mode: Enum = SvdMode.LAPACK
    scaling: Enum = None
    mask_center_px: int = None
    source_xy: Tuple[int] = None
    delta_rot: int = None
    fwhm: float = 4
    adimsdi: Enum = Adimsdi.SINGLE
    crop_ifs: bool = True
    imlib: Enum = Imlib.VIPFFT
    imlib2: Enum = Imlib.VIPFFT
    interpolation: Enum = Interpolation.LANCZOS4
    collapse: Enum = Collapse.MEDIAN
    collapse_ifs: Enum = Collapse.MEAN
    ifs_collapse_range: Union[str, Tuple[int]] = "all"
    smooth: float = None
    smooth_first_pass: float = None
    mask_rdi: np.ndarray = None
    ref_strategy: str = "RDI"
    check_memory: bool = True
    batch: Union[int, float] = None
    nproc: int = 1
    full_output: bool = False
    verbose: bool = True
    weights: np.ndarray = None
    left_eigv: bool = False
    min_frames_pca: int = 10
    max_frames_pca: int = None
    cube_sig: np.ndarray = None
    med_of_npcs: bool = False


def _s(x):
    return str(getattr(x, "value", x)) if x is not None else None


def _is_array(x):
    return isinstance(x, np.ndarray) or B.is_device_tensor(x)


def _project_subtract(cube_t, cube_ref_t, ncomp, scaling, mask_center_px, svd_mode, verbose, full_output):
    """Device version of the whole-matrix branch of the reference's ``_project_subtract``.
    cube_t / cube_ref_t: float32 cuda tensors (n, y, x).  Returns device tensors."""
    n, y, x = cube_t.shape
    if not isinstance(ncomp, (int, np.integer, float, np.floating)):
        raise TypeError("Type not recognized for ncomp, should be int or float")
    if isinstance(ncomp, (float, np.floating)):
        if not 1 > ncomp > 0:
            raise ValueError("if `ncomp` is float, it must lie in the interval (0,1]")
        dec = SVDecomposer(cube_t, mode="fullfr", svd_mode=svd_mode, scaling=scaling, verbose=False)
        ncomp = dec.cevr_to_ncomp(float(ncomp))
        if verbose:
            print("Components used : {}".format(ncomp))
    ncomp = int(ncomp)
    mask = None
    if mask_center_px:
        mask = B.to_device_f32(center_mask_u8((y, x), mask_center_px).astype(np.float32)).to(B._torch().uint8)

    def prep(c):
        m = c.reshape(c.shape[0], -1)
        if mask is not None:
            m = B.apply_mask(m, mask.reshape(-1), 0.0)
        if scaling is not None:
            m = B.scale(m, scaling)
        return m

    M = prep(cube_t)
    ref = prep(cube_ref_t) if cube_ref_t is not None else None
    nref = M.shape[0] if ref is None else ref.shape[0]
    if ncomp > min(nref, M.shape[1]):
        msg = "{} PCs cannot be obtained from a matrix with size [{},{}]."
        msg += " Increase the size of the patches or request less PCs"
        raise RuntimeError(msg.format(ncomp, nref, M.shape[1]))
    res, recon, pcs, _ = B.pca_project(M, ncomp, ref=ref, want_recon=full_output, want_pcs=full_output)
    if verbose:
        print("Done PCA on MI355X (Gram + block-Jacobi + MFMA projection)")
    res = res.reshape(n, y, x)
    if full_output:
        return res, recon, pcs
    return res


def _adi_rdi_pca(cube, cube_ref, angle_list, ncomp, batch, source_xy, delta_rot, fwhm, scaling,
                 mask_center_px, svd_mode, imlib, interpolation, collapse, verbose, start_time, nproc,
                 full_output, weights=None, mask_rdi=None, cube_sig=None, left_eigv=False,
                 min_frames_pca=10, max_frames_pca=None, smooth=None, **rot_options):
    """ADI / ADI+RDI full-frame PCA on device tensors; returns device tensors."""
    if batch is not None:
        raise NotImplementedError("batch (incremental PCA) is outside the accelerated path")
    if source_xy is not None:
        raise NotImplementedError("source_xy (PA-threshold frame rejection) is not accelerated yet")
    if mask_rdi is not None or cube_sig is not None or left_eigv or smooth is not None:
        raise NotImplementedError("mask_rdi / cube_sig / left_eigv / smooth are outside the accelerated path")
    if _s(imlib) != "vip-fft":
        raise NotImplementedError("vip_amd implements imlib='vip-fft' only")
    n, y, x = cube.shape
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=np.float64))
    if not n == angle_list.shape[0]:
        raise ValueError("`angle_list` vector has wrong length. It must equal the number of frames in the cube")
    if not np.isscalar(ncomp) and not isinstance(ncomp, (tuple, list)):
        raise TypeError("`ncomp` must be an int, float, tuple or list in the ADI case")
    if not np.isscalar(ncomp):
        raise NotImplementedError("tuple/list ncomp (pca_grid) is not accelerated yet")
    nref = cube_ref.shape[0] if cube_ref is not None else n
    if isinstance(ncomp, (int, np.integer)) and ncomp > nref:
        ncomp = min(int(ncomp), nref)
        print("Number of PCs too high (max PCs={}), using {} PCs instead.".format(nref, ncomp))
    elif ncomp <= 0:
        raise ValueError("Number of PCs too low. It should be > 0.")
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    if not mv_nan and mask_val != 0:
        raise NotImplementedError("mask_val must be np.nan or 0 on the device path")
    if rot_options.get("edge_blend") not in (None, ""):
        raise NotImplementedError("edge_blend is outside the accelerated path")
    scaling = _s(scaling)
    collapse = _s(collapse)
    if collapse not in B.COLLAPSE_MODES:
        raise TypeError("mode not recognized")
    if collapse == "wmean" and weights is None:
        raise ValueError("Weights have to be provided for weighted mean mode")

    fused_ok = (cube_ref is None and isinstance(ncomp, (int, np.integer)) and collapse in
                ("median", "mean", "sum", "max", "absmean") and (bool(mask_center_px) != mv_nan))
    if fused_ok:
        # one call into the C ABI: mask/scale -> Gram -> eigh -> project -> derotate -> collapse
        mask = None
        if mask_center_px:
            mask = B.to_device_f32(center_mask_u8((y, x), mask_center_px).astype(np.float32)).to(B._torch().uint8)
        out = B.pca_fullframe(cube, angle_list, int(ncomp), scaling=scaling, mask_u8=mask,
                              collapse_mode=collapse, full_output=full_output)
        if verbose:
            print("Done PCA, de-rotating and combining on MI355X")
        if full_output:
            frame, pcs, recon, residuals_cube, residuals_cube_ = out
            return pcs, recon, residuals_cube, residuals_cube_, frame
        return out

    res = _project_subtract(cube, cube_ref, ncomp, scaling, mask_center_px, svd_mode, verbose, full_output)
    if full_output:
        residuals_cube, recon, pcs = res
        pcs = pcs.reshape(pcs.shape[0], y, x)
        recon = recon.reshape(n, y, x)
    else:
        residuals_cube = res
    residuals_cube_ = B.derotate(residuals_cube, angle_list, mask_nan=mv_nan, mask_zero=not mv_nan)
    frame = B.collapse(residuals_cube_, collapse, w=weights)
    if mask_center_px:
        mask = B.to_device_f32(center_mask_u8((y, x), mask_center_px).astype(np.float32)).to(B._torch().uint8)
        if full_output:
            residuals_cube_ = B.apply_mask(residuals_cube_.reshape(n, -1), mask.reshape(-1), 0.0).reshape(n, y, x)
        frame = B.apply_mask(frame.reshape(1, -1), mask.reshape(-1), 0.0).reshape(y, x)
    if verbose:
        print("Done de-rotating and combining")
    if full_output:
        return pcs, recon, residuals_cube, residuals_cube_, frame
    return frame


def pca(*all_args: List, **all_kwargs: dict):
    """Full-frame PCA (ADI, ADI+RDI, 4-D per-channel) on the MI355X.  See the reference docstring
    (psfsub/pca_fullfr.py:137-395) for the meaning of every parameter; returns

    * ``frame`` (``full_output=False``), or
    * ``(frame, pcs, recon, residuals_cube, residuals_cube_)`` for 3-D cubes, with
      ``ifs_adi_frames`` appended and a leading channel axis on the cubes for 4-D input.
    """
    class_params, rot_options = separate_kwargs_dict(initial_kwargs=all_kwargs, parent_class=PCA_Params)
    algo_params = None
    if ALGO_KEY in rot_options.keys():
        algo_params = rot_options[ALGO_KEY]
        del rot_options[ALGO_KEY]
    if algo_params is None:
        algo_params = PCA_Params(*all_args, **class_params)

    # by default, interpolate masked area before derotation if a mask is used (pca_fullfr.py:412-415)
    if algo_params.mask_center_px and len(rot_options) == 0:
        rot_options["mask_val"] = 0
        rot_options["ker"] = 1
        rot_options["interp_zeros"] = True

    cube = algo_params.cube
    if algo_params.batch is not None:
        raise NotImplementedError("batch (incremental PCA) is outside the accelerated path")
    if not _is_array(cube):
        raise TypeError("`cube` must be a 3 or 4d numpy ndarray")
    if cube.ndim not in (3, 4):
        raise TypeError("`cube` must be a 3 or 4d numpy ndarray")
    if algo_params.left_eigv:
        raise NotImplementedError("left_eigv is outside the accelerated path")
    if algo_params.scale_list is not None:
        raise NotImplementedError("scale_list (ADI+mSDI) is not accelerated yet (SURVEY 8(f))")
    if _s(algo_params.svd_mode) not in SVD_MODES:
        raise ValueError("The SVD `mode` is not recognized")
    cond_mask = algo_params.mask_rdi is not None
    if cond_mask and algo_params.ref_strategy in ("ARDI", "ARSDI"):
        raise TypeError("mask for data imputation detected. This mode can only run with a pure RDI strategy, "
                        "while ref_strategy was set to {}".format(algo_params.ref_strategy))

    dev_in = B.is_device_tensor(cube)
    out_dtype = None if dev_in else (cube.dtype if cube.dtype.kind == "f" else np.float64)
    if algo_params.check_memory:
        torch = B.require_gpu()
        free, _total = torch.cuda.mem_get_info()
        need = int(np.prod(cube.shape)) * 4 * 8
        if need > free:
            raise RuntimeError("Input cube needs ~{:.1f} GB of HBM ({:.1f} GB free). Set check_memory=False "
                               "to override".format(need / 1e9, free / 1e9))

    def host(t, dtype=None):
        if dev_in:
            return t
        return t.cpu().numpy().astype(dtype or out_dtype, copy=False)

    cube_t = B.to_device_f32(cube)
    cube_ref_t = None
    if algo_params.cube_ref is not None:
        cube_ref_t = B.to_device_f32(algo_params.cube_ref)

    fo = bool(algo_params.full_output)
    add = {"start_time": None, "full_output": fo}

    if cube.ndim == 4:
        torch = B._torch()
        nch, nz, ny, nx = cube.shape
        ncomp = algo_params.ncomp
        if not isinstance(ncomp, list):
            ncomps = [ncomp] * nch
        elif len(ncomp) != nch:
            raise NotImplementedError("list ncomp of length != n_channels (pca_grid) is not accelerated yet")
        else:
            ncomps = ncomp
        fwhm = algo_params.fwhm
        fwhms = [fwhm] * nch if np.isscalar(fwhm) else fwhm
        outs = []
        for ch in range(nch):
            ref_ch = None
            if cube_ref_t is not None:
                if cube_ref_t.ndim != 4:
                    raise TypeError("Ref cube has wrong format for 4d input cube")
                if algo_params.ref_strategy == "RDI":
                    ref_ch = cube_ref_t[ch]
                elif algo_params.ref_strategy == "ARDI":
                    ref_ch = torch.cat((cube_t[ch], cube_ref_t[ch]))
                else:
                    raise TypeError("ref_strategy argument not recognized.Should be 'RDI' or 'ARDI'")
            fp = setup_parameters(algo_params, _adi_rdi_pca, cube=cube_t[ch], cube_ref=ref_ch,
                                  ncomp=ncomps[ch], fwhm=fwhms[ch], **add)
            outs.append(_adi_rdi_pca(**fp, **rot_options))
        ifs = torch.stack([o[4] if fo else o for o in outs])
        frame = B.collapse(ifs, _s(algo_params.collapse_ifs))
        if fo:
            pcs = torch.stack([o[0] for o in outs])
            recon = torch.stack([o[1] for o in outs])
            res = torch.stack([o[2] for o in outs])
            resd = torch.stack([o[3] for o in outs])
            # reference dtypes: frame / ifs_adi_frames float64, cubes float32 (pca_fullfr.py:546)
            return (host(frame, np.float64), host(pcs), host(recon), host(res), host(resd),
                    host(ifs, np.float64))
        return host(frame, np.float64)

    # 3-D ADI / RDI
    if cube_ref_t is not None:
        if cube_ref_t.ndim != 3:
            raise TypeError("Ref cube has wrong format for 3d input cube")
        if algo_params.ref_strategy == "ARDI":
            cube_ref_t = B._torch().cat((cube_t, cube_ref_t))
        elif algo_params.ref_strategy != "RDI":
            raise TypeError("ref_strategy argument not recognized.Should be 'RDI' or 'ARDI'")
    fp = setup_parameters(algo_params, _adi_rdi_pca, cube=cube_t, cube_ref=cube_ref_t, **add)
    out = _adi_rdi_pca(**fp, **rot_options)
    if fo:
        pcs, recon, residuals_cube, residuals_cube_, frame = out
        return host(frame), host(pcs), host(recon), host(residuals_cube), host(residuals_cube_)
    return host(out)
