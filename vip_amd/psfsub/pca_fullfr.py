"""Full-frame PCA PSF subtraction: drop-in for ``vip_hci.psfsub.pca`` (reference
psfsub/pca_fullfr.py:93-134 PCA_Params, :137-798 pca, :801-1035 _adi_rdi_pca, :1552-1737
_project_subtract) for the ADI / RDI (3-D) and per-channel (4-D, no ``scale_list``) branches.

Same positional order (= ``PCA_Params`` field order), same kwargs (unknown kwargs become
``rot_options``), same return tuples and shapes.  All array work runs on the MI355X through
libvipmi.so; numpy in -> numpy out, cuda tensor in -> cuda tensors out (no host copies).

Tuple/list ``ncomp`` (the ``pca_grid`` of final frames, with S/N scoring at ``source_xy``), ``source_xy`` (PA-threshold
frame rejection), ``cube_ref`` (RDI / ARDI), ``cube_sig`` and 4-D cubes with ``scale_list`` (ADI+mSDI, psfsub/pca_msdi.py)
are accelerated, ``left_eigv`` for plain ADI.  Not accelerated (raise NotImplementedError): ``batch`` (incremental PCA), ``mask_rdi``,
``smooth``, ``imlib`` other than 'vip-fft' (parity path) and 'opencv' (interpolating rotation, 3-D cubes).
"""
from dataclasses import dataclass
from enum import Enum
from typing import List, Tuple, Union

import os

import numpy as np

from .. import backend as B
from ..config.paramenum import ALGO_KEY, Adimsdi, Collapse, Imlib, Interpolation, SvdMode
from ..config.utils_param import separate_kwargs_dict, setup_parameters
from ..preproc.derotation import _compute_pa_thresh, _find_indices_adi_all
from ..preproc.parangles import check_pa_vector
from ..var.coords import dist, frame_center
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


def _project_subtract(cube_t, cube_ref_t, ncomp, scaling, mask_center_px, svd_mode, verbose, full_output,
                      cube_sig_t=None):
    """Device version of the whole-matrix branch of the reference's ``_project_subtract``.
    cube_t / cube_ref_t / cube_sig_t: float32 cuda tensors (n, y, x).  Returns device tensors.

    ``cube_sig`` (reference :1652-1662,1717-1731): PCs and projection come from the "empty" matrix
    ``M_emp = M - S`` (S = cube_sig, neither masked nor scaled) while the model is subtracted from ``M``; since
    ``M - proj(M_emp) = (M_emp - proj(M_emp)) + S`` this is the ordinary path on ``M_emp`` plus ``S`` added back."""
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
    S = None
    if cube_sig_t is not None:
        S = cube_sig_t.reshape(cube_sig_t.shape[0], -1)
        M = B.lincomb(M, S, 1.0, -1.0)
    ref = prep(cube_ref_t) if cube_ref_t is not None else None
    nref = M.shape[0] if ref is None else ref.shape[0]
    if ncomp > min(nref, M.shape[1]):
        msg = "{} PCs cannot be obtained from a matrix with size [{},{}]."
        msg += " Increase the size of the patches or request less PCs"
        raise RuntimeError(msg.format(ncomp, nref, M.shape[1]))
    res, recon, pcs, _ = B.pca_project(M, ncomp, ref=ref, want_recon=full_output, want_pcs=full_output)
    if S is not None:
        res = B.lincomb(res, S, 1.0, 1.0)
    if verbose:
        print("Done PCA on MI355X (Gram + eigensolver + MFMA projection)")
    res = res.reshape(n, y, x)
    if full_output:
        return res, recon, pcs
    return res


def _prepared_matrices(cube_t, cube_ref_t, scaling, mask_center_px):
    """prepare_matrix(mode='fullfr') of the cube (and reference cube) on the device -> (M, ref or None)."""
    n, y, x = cube_t.shape
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

    return prep(cube_t), (prep(cube_ref_t) if cube_ref_t is not None else None)


def _pca_pa_rejection(cube, angle_list, ncomp, source_xy, delta_rot, fwhm, scaling, mask_center_px, min_frames_pca,
                      max_frames_pca, verbose, cube_sig=None, cube_ref=None):
    """Device version of the ``source_xy`` branch (reference pca_fullfr.py:911-965 + the per-frame mode of
    ``_project_subtract``, :1677-1713): frame j is modelled with the PCs of the frames that have rotated by more than
    the PA threshold at ``source_xy`` -- plus, with ``cube_ref``, every reference frame (:1693-1694).  All n per-frame
    decompositions come from sub-blocks of ONE Gram matrix (the identity used for annular PCA).
    Returns (residuals (n, P), M (n, P), library sizes)."""
    torch = B._torch()
    n, y, x = cube.shape
    if delta_rot is None or fwhm is None:
        raise TypeError("Delta_rot or fwhm parameters missing. Needed forPA-based rejection of frames from the library")
    yc, xc = frame_center(cube[0])
    x1, y1 = source_xy
    ann_center = dist(yc, xc, y1, x1)
    pa_thr = _compute_pa_thresh(ann_center, fwhm, delta_rot)
    truncate = max_frames_pca is not None
    libs = _find_indices_adi_all(angle_list, pa_thr, truncate=truncate, max_frames=max_frames_pca)
    nr = 0 if cube_ref is None else int(cube_ref.shape[0])
    msg = "{} frames comply to delta_rot condition < less than "
    for li in libs:
        if li.shape[0] + nr < min_frames_pca:
            raise RuntimeError((msg + "min_frames_pca ({}). Try decreasing delta_rot or min_frames_pca").format(
                li.shape[0] + nr, min_frames_pca))
        if li.shape[0] + nr < ncomp:
            raise RuntimeError((msg + "ncomp ({}). Try decreasing the parameter delta_rot or ncomp").format(
                li.shape[0] + nr, ncomp))
    M, Mref = _prepared_matrices(cube, cube_ref, scaling, mask_center_px)
    S = None
    if cube_sig is not None:                      # libraries and projections from M - S (see _project_subtract)
        S = cube_sig.reshape(n, -1)
        M_full, M = M, B.lincomb(M, S, 1.0, -1.0)
    P = y * x
    A = M if nr == 0 else torch.cat((M, Mref)).contiguous()        # rows n .. n + nr - 1: the reference frames
    ntot = n + nr
    max_lib = max(li.shape[0] for li in libs) + nr
    idx = np.zeros((ntot, max_lib), dtype=np.int32)
    ln = np.zeros(ntot, dtype=np.int32)
    refrows = np.arange(n, ntot, dtype=np.int32)
    for j, li in enumerate(libs):
        m = li.shape[0] + nr
        idx[j, :li.shape[0]] = li
        idx[j, li.shape[0]:m] = refrows
        ln[j] = m
    idx[n:] = idx[0]                              # (the reference rows need some valid library; their residuals are dropped)
    ln[n:] = ln[0]
    idx_t = torch.from_numpy(idx).to(cube.device)
    ln_t = torch.from_numpy(ln).to(cube.device)
    R = B.empty((ntot, P), device=cube.device.index)
    ctx = B.get_context(cube.device.index)
    ctx.call("vipmi_annular_residuals_f32", B.ptr(A), ntot, P, B.ptr(idx_t), B.ptr(ln_t), int(max_lib), int(ncomp), B.ptr(R))
    R = R[:n]
    if S is not None:
        R = B.lincomb(R.contiguous(), S, 1.0, 1.0)
        M = M_full
    if verbose:
        print("Size LIB: min={} max={} mean={:.1f}".format(int(ln[:n].min()), int(ln[:n].max()), float(ln[:n].mean())))
    return R, M, ln[:n]


@B.with_rotation
def _adi_rdi_pca(cube, cube_ref, angle_list, ncomp, batch, source_xy, delta_rot, fwhm, scaling,
                 mask_center_px, svd_mode, imlib, interpolation, collapse, verbose, start_time, nproc,
                 full_output, weights=None, mask_rdi=None, cube_sig=None, left_eigv=False,
                 min_frames_pca=10, max_frames_pca=None, smooth=None, **rot_options):
    """ADI / ADI+RDI full-frame PCA on device tensors; returns device tensors."""
    if batch is not None:
        raise NotImplementedError("batch (incremental PCA) is outside the accelerated path")
    if mask_rdi is not None or smooth is not None:
        raise NotImplementedError("mask_rdi / smooth are outside the accelerated path")
    if left_eigv and (cube_ref is not None or source_xy is not None or not isinstance(ncomp, (int, np.integer))):
        raise NotImplementedError("left_eigv: plain ADI with an integer ncomp only")
    if cube_sig is not None and tuple(cube_sig.shape) != tuple(cube.shape):
        raise TypeError("`cube_sig` must have the shape of `cube`")
    B.check_imlib(imlib, interpolation)         # 'vip-fft' or 'opencv'; the decorator selects the rotation
    n, y, x = cube.shape
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=np.float64))
    if not n == angle_list.shape[0]:
        raise ValueError("`angle_list` vector has wrong length. It must equal the number of frames in the cube")
    if not np.isscalar(ncomp) and not isinstance(ncomp, (tuple, list)):
        raise TypeError("`ncomp` must be an int, float, tuple or list in the ADI case")
    grid = not np.isscalar(ncomp)
    nref = cube_ref.shape[0] if cube_ref is not None else n
    if grid:
        pass
    elif isinstance(ncomp, (int, np.integer)) and ncomp > nref:
        ncomp = min(int(ncomp), nref)
        print("Number of PCs too high (max PCs={}), using {} PCs instead.".format(nref, ncomp))
    elif ncomp <= 0:
        raise ValueError("Number of PCs too low. It should be > 0.")
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    mv_other = B.other_mask_value(mask_val) is not None     # neither NaN nor 0: B.derotate takes it from the rotation scope
    if rot_options.get("edge_blend") not in (None, ""):
        raise NotImplementedError("edge_blend is outside the accelerated path")
    scaling = _s(scaling)
    collapse = _s(collapse)
    if collapse not in B.COLLAPSE_MODES or collapse == "stim":        # ('stim' is an internal mode, subsampling.py:79-114)
        raise TypeError("mode not recognized")
    if collapse == "wmean" and weights is None:
        raise ValueError("Weights have to be provided for weighted mean mode")

    if grid:
        # pca_fullfr.py:1010-1035: one decomposition, every truncation derotated + collapsed on the device; with
        # source_xy the frames are scored by the mean S/N in a FWHM aperture (host) -> (cubeout, finalfr, df, opt_npc)
        if cube_sig is not None:
            raise NotImplementedError("cube_sig with a grid of ncomp is outside the accelerated path")
        from .utils_pca import pca_grid
        return pca_grid(cube, angle_list, fwhm, range_pcs=ncomp, source_xy=source_xy, cube_ref=cube_ref, mode="fullfr",
                        svd_mode=svd_mode, scaling=scaling, mask_center_px=mask_center_px, fmerit="mean",
                        collapse=collapse, verbose=verbose, full_output=full_output, debug=False, plot=False,
                        weights=weights, imlib=imlib, interpolation=interpolation, **rot_options)
    if source_xy is not None:
        if not isinstance(ncomp, (int, np.integer)):
            raise NotImplementedError("source_xy needs an integer ncomp on the device path")
        R, M, _ln = _pca_pa_rejection(cube, angle_list, int(ncomp), source_xy, delta_rot, fwhm, scaling,
                                      mask_center_px, min_frames_pca, max_frames_pca, verbose, cube_sig=cube_sig,
                                      cube_ref=cube_ref)
        residuals_cube = R.reshape(n, y, x)
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
            recon_cube = B.lincomb(M, R, 1.0, -1.0).reshape(n, y, x)
            return recon_cube, residuals_cube, residuals_cube_, frame
        return frame

    def left_pcs():
        """pcs of ``left_eigv=True``: the temporal modes as rows, (ncomp x n) (pca_fullfr.py:905) -- the residuals are
        those of the ordinary projection, U U^T M = M V^T V for the leading k triplets of M itself."""
        from .svd import svd_wrapper
        m = cube.reshape(n, -1)
        if mask_center_px:
            mk = B.to_device_f32(center_mask_u8((y, x), mask_center_px).astype(np.float32)).to(B._torch().uint8)
            m = B.apply_mask(m, mk.reshape(-1), 0.0)
        if scaling is not None:
            m = B.scale(m, scaling)
        if cube_sig is not None:
            m = B.lincomb(m, cube_sig.reshape(n, -1), 1.0, -1.0)
        return svd_wrapper(m, svd_mode, min(int(ncomp), n), False, left_eigv=True).t().contiguous()

    fused_ok = (cube_ref is None and cube_sig is None and n <= B.MAX_EIGH_N and _s(imlib) == "vip-fft" and isinstance(ncomp, (int, np.integer)) and collapse in
                ("median", "mean", "sum", "max", "absmean") and (bool(mask_center_px) != mv_nan) and not mv_other)
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
            if left_eigv:
                pcs = left_pcs()
            return pcs, recon, residuals_cube, residuals_cube_, frame
        return out

    res = _project_subtract(cube, cube_ref, ncomp, scaling, mask_center_px, svd_mode, verbose, full_output,
                            cube_sig_t=cube_sig)
    if full_output:
        residuals_cube, recon, pcs = res
        pcs = left_pcs() if left_eigv else pcs.reshape(pcs.shape[0], y, x)
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


def _adi_pca_channels_batched(cube4, angle_list, ncomp, scaling, mask_center_px, collapse, weights, mv_nan):
    """Plain ADI PCA of every spectral channel of a 4-D cube (same integer ``ncomp``, no reference cube, final frames
    only) with the small per-channel stages batched: ONE Gram launch and ONE eigensolver launch for all channels (a
    workgroup per channel), ONE derotation call over all nch * n residual frames; only the projection products and the
    collapses stay per channel.  Same arithmetic as the per-channel loop of ``pca`` (reference pca_fullfr.py:544-658).
    Returns the per-channel final frames (nch, y, x) as a device tensor."""
    torch = B._torch()
    nch, n, y, x = cube4.shape
    P = y * x
    k = int(ncomp)
    mask = None
    if mask_center_px:
        mask = B.to_device_f32(center_mask_u8((y, x), mask_center_px).astype(np.float32)).to(torch.uint8)
    if mask is None and scaling is None:
        M = cube4.reshape(nch, n, P)
    else:
        mats = []
        for c in range(nch):
            m = cube4[c].reshape(n, P)
            if mask is not None:
                m = B.apply_mask(m, mask.reshape(-1), 0.0)
            if scaling is not None:
                m = B.scale(m, scaling)
            mats.append(m)
        M = torch.stack(mats)
    G = B.gram_batched(M)
    ev, ec = B.eigh_topk(G, k)                                            # (nch, k), (nch, k, n)
    keep = (ev > ev[:, :1] * 1e-12).to(torch.float32)
    E = (ec.to(torch.float32) * keep[:, :, None]).contiguous()
    R = B.project_batched(M.contiguous(), E)                              # every channel's projection in two launches
    der = B.derotate(R.reshape(nch * n, y, x), np.tile(angle_list, nch), mask_nan=mv_nan, mask_zero=not mv_nan)
    der = der.reshape(nch, n, y, x)
    frames = B.collapse_batched(der, collapse, w=weights)                 # every channel in one launch
    if mask is not None:
        frames = B.apply_mask(frames.reshape(nch, -1), mask.reshape(-1), 0.0).reshape(nch, y, x)
    return frames


def _float64_fused(algo_params, rot_options, cube):
    """Plain 3-D ADI PCA of a FLOAT64 cube: the fused float64 entry (csrc/pca_f64.hip), which carries the
    per-pixel temporal mean in float64 -- the reference keeps the caller's dtype through svd_wrapper (pca_fullfr.py:1552-1737), and
    rounding a cube of detector counts to float32 first costs 2e-3 on the final frame (golden g28).  Returns the frame -- (frame, pcs, recon, residuals, residuals_der) with full_output -- as cuda
    tensors, or None when the call is not of that shape (everything else converts to float32 as before)."""
    torch = B._torch() if B.is_device_tensor(cube) else None
    is64 = (cube.dtype == np.float64) if torch is None else (cube.dtype == torch.float64)
    ap = algo_params
    if not is64 or cube.ndim != 3 or ap.left_eigv:
        return None
    if any(getattr(ap, name, None) is not None for name in ("cube_ref", "cube_sig", "scale_list", "source_xy", "batch", "mask_rdi", "smooth")):
        return None
    if not isinstance(ap.ncomp, (int, np.integer)) or isinstance(ap.ncomp, bool) or ap.ncomp <= 0:
        return None
    scaling, collapse = _s(ap.scaling), _s(ap.collapse)
    if scaling not in B.SCALE_MODES or collapse not in ("median", "mean", "sum", "max", "absmean"):
        return None
    if _s(ap.imlib) != "vip-fft" or _s(ap.svd_mode) not in SVD_MODES or rot_options.get("edge_blend") not in (None, ""):
        return None
    n, y, x = cube.shape
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    if y != x or n > B.MAX_EIGH_N or B.other_mask_value(mask_val) is not None or (bool(ap.mask_center_px) == mv_nan):
        return None
    angle_list = check_pa_vector(np.asarray(ap.angle_list, dtype=np.float64))
    if angle_list.shape[0] != n:
        raise ValueError("`angle_list` vector has wrong length. It must equal the number of frames in the cube")
    ncomp = int(ap.ncomp)
    if ncomp > n:
        ncomp = n
        print("Number of PCs too high (max PCs={}), using {} PCs instead.".format(n, ncomp))
    torch = B.require_gpu()
    if B.is_device_tensor(cube):
        c64 = cube.to(device=torch.device("cuda", torch.cuda.current_device()))
    else:
        c64 = torch.from_numpy(np.ascontiguousarray(cube)).to(torch.device("cuda", torch.cuda.current_device()))
    mask = None
    if ap.mask_center_px:
        mask = B.to_device_f32(center_mask_u8((y, x), ap.mask_center_px).astype(np.float32)).to(torch.uint8)
    out = B.pca_fullframe_f64(c64, angle_list, ncomp, scaling=scaling, mask_u8=mask, collapse_mode=collapse,
                              full_output=bool(ap.full_output))
    if ap.verbose:
        print("Done PCA (float64 cube: temporal mean carried in float64), de-rotating and combining on MI355X")
    return out


def _hostin_fused(algo_params, rot_options, cube):
    """Plain 3-D ADI PCA of a float32 NUMPY cube (what the reference's callers pass, pca_fullfr.py:137): the fused entry that
    uploads the cube itself, in blocks of 64 frames, and forms the Gram matrix of the blocks that have arrived while the next one
    is on the link (vipmi_pca_fullframe_hostin_f32; bit-identical to uploading first).  Returns cuda tensors like
    ``_float64_fused`` or None when the call is not of that shape."""
    ap = algo_params
    if B.is_device_tensor(cube) or not isinstance(cube, np.ndarray) or cube.dtype != np.float32 or cube.ndim != 3 or ap.left_eigv:
        return None
    if os.environ.get("VIPMI_HOSTIN", "1") == "0" or cube.nbytes < (64 << 20):
        return None
    if any(getattr(ap, name, None) is not None for name in ("cube_ref", "cube_sig", "scale_list", "source_xy", "batch", "mask_rdi", "smooth")):
        return None
    if not isinstance(ap.ncomp, (int, np.integer)) or isinstance(ap.ncomp, bool) or ap.ncomp <= 0:
        return None
    if _s(ap.scaling) is not None or _s(ap.collapse) not in ("median", "mean", "sum", "max", "absmean"):
        return None
    if _s(ap.imlib) != "vip-fft" or _s(ap.svd_mode) not in SVD_MODES or rot_options.get("edge_blend") not in (None, ""):
        return None
    n, y, x = cube.shape
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    if y != x or n > B.MAX_EIGH_N or B.other_mask_value(mask_val) is not None or (bool(ap.mask_center_px) == mv_nan):
        return None
    angle_list = check_pa_vector(np.asarray(ap.angle_list, dtype=np.float64))
    if angle_list.shape[0] != n:
        raise ValueError("`angle_list` vector has wrong length. It must equal the number of frames in the cube")
    ncomp = int(ap.ncomp)
    if ncomp > n:
        ncomp = n
        print("Number of PCs too high (max PCs={}), using {} PCs instead.".format(n, ncomp))
    mask = None
    if ap.mask_center_px:
        mask = B.to_device_f32(center_mask_u8((y, x), ap.mask_center_px).astype(np.float32)).to(B._torch().uint8)
    out = B.pca_fullframe_hostin(np.ascontiguousarray(cube), angle_list, ncomp, mask_u8=mask, collapse_mode=_s(ap.collapse),
                                 full_output=bool(ap.full_output))
    if ap.verbose:
        print("Done PCA (Gram matrix formed under the upload), de-rotating and combining on MI355X")
    return out


def _hostin_4d(algo_params, rot_options, cube):
    """4-D float32 NUMPY cube without ``scale_list``, final frame only (pca_fullfr.py:544-658): the channels are uploaded in a few
    groups on a copy stream while the batched per-channel path (``_adi_pca_channels_batched``) works on the group before -- the
    upload of an IFS cube takes longer than its PCA (C4: 36 ms against 24).  Every group's Gram launch is pinned to the split-K
    slicing the whole batch would get, and every other stage is per problem / per frame: bit-identical to uploading first.  None
    when the call is not of that shape."""
    ap = algo_params
    if B.is_device_tensor(cube) or not isinstance(cube, np.ndarray) or cube.dtype != np.float32 or cube.ndim != 4:
        return None
    if os.environ.get("VIPMI_HOSTIN", "1") == "0" or cube.nbytes < (256 << 20) or ap.full_output or ap.left_eigv:
        return None
    if any(getattr(ap, name, None) is not None for name in ("cube_ref", "cube_sig", "scale_list", "source_xy", "batch", "mask_rdi", "smooth")):
        return None
    nch, nz, ny, nx = cube.shape
    ncomp = ap.ncomp
    if not isinstance(ncomp, (int, np.integer)) or isinstance(ncomp, bool) or not (0 < int(ncomp) <= min(64, nz)) or nz > 512:
        return None
    if _s(ap.imlib) != "vip-fft" or _s(ap.svd_mode) not in SVD_MODES or rot_options.get("edge_blend") not in (None, ""):
        return None
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    if ny != nx or not (mv_nan or mask_val == 0) or _s(ap.collapse) not in ("median", "mean", "sum", "max", "absmean"):
        return None
    if _s(ap.collapse_ifs) not in ("median", "mean", "sum", "max", "absmean") or _s(ap.scaling) is not None or ap.ifs_collapse_range != "all":
        return None
    P = ny * nx
    ngrp = min(int(os.environ.get("VIPMI_HOSTIN_GROUPS", "8")), nch // 4)      # (C4: 2 groups 53 ms, 4: 49, 8: 48 median / 43 best; 62 without)
    bounds = [round(g * nch / ngrp) for g in range(ngrp + 1)] if ngrp >= 2 else []
    # the int8 Gram path must be the one every group AND the whole batch would take (gram.hip: gram_batched_f32)
    if ngrp < 2 or nz < 128 or P < 16384 or min(b1 - b0 for b0, b1 in zip(bounds, bounds[1:])) * nz * P < (1 << 25):
        return None
    torch = B.require_gpu()
    if B.is_async():
        return None
    angles = check_pa_vector(np.asarray(ap.angle_list, dtype=np.float64))
    if angles.shape[0] != nz:
        raise ValueError("`angle_list` vector has wrong length. It must equal the number of frames in the cube")
    dev = torch.cuda.current_device()
    cur = torch.cuda.current_stream()
    copy_stream = B.side_streams(1, dev)[0]
    ctx = B.get_context(dev)
    # split-K slices of the whole batch's Gram launch (gram_i8.hip run(): ~6 workgroups per CU over tiles x problems)
    nt = -(-nz // 64)
    want = max(1, -(-6 * torch.cuda.get_device_properties(dev).multi_processor_count // (nt * (nt + 1) // 2 * nch)))
    had = ctx.get_option("gram_i8_slices")
    frames = []
    B.set_async(True)
    try:
        ctx.set_option("gram_i8_slices", want)
        copy_stream.wait_stream(cur)
        # the (blocking, pageable) copies run on a thread of their own: the interpreter lock is released inside the copy, so this
        # thread enqueues the kernels of group g while group g + 1 is on the link (host time per group ~2.5 ms: in one thread the
        # call took upload + 4 x that)
        import queue
        import threading
        q = queue.Queue()

        def uploader():
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(copy_stream):
                    for g in range(ngrp):
                        t_ = torch.from_numpy(np.ascontiguousarray(cube[bounds[g]:bounds[g + 1]])).to(torch.device("cuda", dev))
                        q.put((t_, copy_stream.record_event()))
            except BaseException as e:          # (handed to the consumer: it must not wait for a group that will never come)
                q.put(e)

        th = threading.Thread(target=uploader, name="vipmi-upload", daemon=True)
        th.start()
        try:
            for g in range(ngrp):
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                t, ev = item
                cur.wait_event(ev)
                t.record_stream(cur)
                frames.append(_adi_pca_channels_batched(t, angles, int(ncomp), None, ap.mask_center_px, _s(ap.collapse), ap.weights, mv_nan))
                del t, item
        finally:
            th.join()
        ifs = torch.cat(frames)
        frame = B.collapse(ifs, _s(ap.collapse_ifs))
        cur.synchronize()
        B.check_deferred()
    finally:
        ctx.set_option("gram_i8_slices", had)
        B.set_async(False)
    if ap.verbose:
        print("Done PCA per channel (channel groups uploaded beside the PCA of the group before), combining on MI355X")
    return frame


def _float64_fused_4d(algo_params, rot_options, cube):
    """4-D float64 cube without ``scale_list`` (pca_fullfr.py:544-658), final frame only: every spectral channel through the
    float64 route of ``_float64_fused``, then ``collapse_ifs`` of the per-channel frames.  None when the call is not of that shape."""
    import copy
    ap = algo_params
    torch = B._torch() if B.is_device_tensor(cube) else None
    is64 = (cube.dtype == np.float64) if torch is None else (cube.dtype == torch.float64)
    if not is64 or cube.ndim != 4 or ap.full_output or ap.scale_list is not None:
        return None
    if _s(ap.collapse_ifs) not in ("median", "mean", "sum", "max", "absmean"):
        return None
    nch = cube.shape[0]
    ncomp = ap.ncomp
    ncomps = list(ncomp) if isinstance(ncomp, list) and len(ncomp) == nch else [ncomp] * nch
    frames = []
    # numpy in: the channels are uploaded by a thread of their own on a copy stream, one or two ahead of the channel the fused
    # float64 call is working on (the interpreter lock is released inside the blocking copy and inside the library call): an
    # IFS cube of float64 takes longer to upload than to process (C4 shape: 74 ms against 47)
    feed = None
    if torch is None and os.environ.get("VIPMI_HOSTIN", "1") != "0" and nch > 1 and cube[0].nbytes >= (16 << 20):
        import queue
        import threading
        tt = B.require_gpu()
        dev = tt.cuda.current_device()
        cur = tt.cuda.current_stream()
        copy_stream = B.side_streams(1, dev)[0]
        copy_stream.wait_stream(cur)
        q = queue.Queue(maxsize=2)

        def uploader():
            try:
                tt.cuda.set_device(dev)
                with tt.cuda.stream(copy_stream):
                    for ch in range(nch):
                        t_ = tt.from_numpy(np.ascontiguousarray(cube[ch])).to(tt.device("cuda", dev))
                        q.put((t_, copy_stream.record_event()))
            except BaseException as e:
                q.put(e)

        th = threading.Thread(target=uploader, name="vipmi-upload", daemon=True)
        th.start()

        def feed(ch):
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            t_, ev = item
            cur.wait_event(ev)
            t_.record_stream(cur)
            return t_
    try:
        for ch in range(nch):
            apc = copy.copy(ap)
            src = feed(ch) if feed is not None else cube[ch]
            apc.cube, apc.ncomp = src, ncomps[ch]
            fr = _float64_fused(apc, rot_options, src)
            if fr is None:
                frames = None                # (some channel is not a plain integer-ncomp ADI call: the float32 route for all)
                break
            frames.append(fr)
    finally:
        if feed is not None:
            while th.is_alive():             # (drain: the uploader may be blocked on a full queue after an early exit)
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()
    if frames is None:
        return None
    return B.collapse(B._torch().stack(frames), _s(ap.collapse_ifs))


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
        if algo_params.batch is not None or algo_params.mask_rdi is not None or algo_params.cube_ref is not None:
            raise NotImplementedError("left_eigv is not compatible with 'mask_rdi' nor 'batch'")      # (pca_fullfr.py:428-437)
        if algo_params.scale_list is not None or cube.ndim == 4:
            raise NotImplementedError("left_eigv with 4-D cubes is outside the accelerated path")
    if _s(algo_params.svd_mode) not in SVD_MODES:
        raise ValueError("The SVD `mode` is not recognized")
    if algo_params.scale_list is not None:
        # argument checks of the ADI+mSDI path before anything touches the GPU
        if cube.ndim != 4:
            raise TypeError("`scale_list` needs a 4d (channels, frames, y, x) cube")
        for name in ("mask_rdi", "cube_sig", "smooth_first_pass"):
            if name == "cube_sig" and _s(algo_params.adimsdi) == "double":
                continue                      # double pass: handed to the second (ADI) stage
            if getattr(algo_params, name, None) is not None:
                raise NotImplementedError("{} is outside the accelerated ADI+mSDI path".format(name))
        if algo_params.cube_ref is not None and np.ndim(algo_params.cube_ref) != 4:
            raise TypeError("Ref cube has wrong format for 4d input cube")
        if _s(algo_params.imlib) != "vip-fft" or _s(algo_params.imlib2) != "vip-fft":
            raise NotImplementedError("vip_amd implements imlib='vip-fft' / imlib2='vip-fft' only")
        if _s(algo_params.adimsdi) not in ("double", "single"):
            raise ValueError("`adimsdi` mode not recognized")
        if _s(algo_params.adimsdi) == "double" and not isinstance(algo_params.ncomp, tuple):
            raise TypeError("`ncomp` must be a tuple when a double pass PCA is performed")
        if np.asarray(algo_params.scale_list).ndim > 1:
            raise ValueError("Scaling factors vector is not 1d")
        if np.asarray(algo_params.scale_list).shape[0] != cube.shape[0]:
            raise ValueError("Scaling factors vector has wrong length")
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
        return B.to_host(t, dtype or out_dtype)

    out64 = _float64_fused(algo_params, rot_options, cube)
    if out64 is None:
        out64 = _hostin_fused(algo_params, rot_options, cube)
    if out64 is not None:
        if algo_params.full_output and not dev_in:
            return tuple(B.to_host_many(list(out64), [out_dtype] * len(out64)))       # (all copies enqueued, one synchronisation)
        return tuple(host(t) for t in out64) if algo_params.full_output else host(out64)
    out64 = _float64_fused_4d(algo_params, rot_options, cube)
    if out64 is None:
        out64 = _hostin_4d(algo_params, rot_options, cube)
    if out64 is not None:
        return host(out64, np.float64)
    cube_t = B.to_device_f32(cube)
    if algo_params.scale_list is not None:
        # ADI+mSDI (pca_fullfr.py:478-540): 4-D cube, channels rescaled by scale_list
        from .pca_msdi import adimsdi_double, adimsdi_single
        mask_val = rot_options.get("mask_val", np.nan)
        mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
        # (the rotation scope carries a mask_val that is neither NaN nor 0 to every B.derotate of the mSDI passes)
        with B.rotation_mode("vip-fft", "lanczos4", "constant", mask_val):
            collapse = _s(algo_params.collapse)
            if collapse not in B.COLLAPSE_MODES or collapse == "stim":
                raise TypeError("mode not recognized")
            mode = _s(algo_params.adimsdi)
            if mode == "double":
                rcc, rcc_, frame = adimsdi_double(cube_t, algo_params.angle_list, algo_params.scale_list,
                                                  algo_params.ncomp, algo_params.scaling, algo_params.mask_center_px,
                                                  collapse, algo_params.collapse_ifs, algo_params.ifs_collapse_range,
                                                  algo_params.weights, mv_nan, algo_params.verbose,
                                                  cube_ref=(None if algo_params.cube_ref is None
                                                            else B.to_device_f32(algo_params.cube_ref)),
                                                  ref_strategy=_s(algo_params.ref_strategy),
                                                  source_xy=algo_params.source_xy, delta_rot=algo_params.delta_rot,
                                                  fwhm=algo_params.fwhm, min_frames_pca=algo_params.min_frames_pca,
                                                  max_frames_pca=algo_params.max_frames_pca,
                                                  cube_sig=(None if algo_params.cube_sig is None
                                                            else B.to_device_f32(algo_params.cube_sig)))
                # the reference's scale_fft returns float32 when it crops the spectrum (down-scaling) and float64 when it
                # pads it or leaves a channel untouched (scale 1): mirror the resulting dtype of the collapsed frames
                dt = None
                if algo_params.ncomp[0] is not None:
                    z_ = cube.shape[0]
                    r0, r1 = (0, z_) if algo_params.ifs_collapse_range == "all" else algo_params.ifs_collapse_range
                    sl = np.asarray(algo_params.scale_list, dtype=np.float64)[r0:r1]
                    dt = np.float64 if np.any(sl <= 1) else np.float32
                if algo_params.full_output:
                    return host(frame, dt), host(rcc, dt), host(rcc_, dt)
                return host(frame, dt)
            if mode == "single":
                ref_t = None
                if algo_params.cube_ref is not None:
                    ref_t = B.to_device_f32(algo_params.cube_ref)
                    if "A" in _s(algo_params.ref_strategy):          # e.g. 'ARSDI': the science frames join the library
                        ref_t = B._torch().cat((cube_t, ref_t), dim=1)
                grid = isinstance(algo_params.ncomp, (tuple, list))
                res = adimsdi_single(cube_t, algo_params.angle_list, algo_params.scale_list,
                                     algo_params.ncomp, _s(algo_params.scaling),
                                     algo_params.mask_center_px, collapse, algo_params.collapse_ifs,
                                     algo_params.ifs_collapse_range, algo_params.crop_ifs,
                                     algo_params.weights, mv_nan, algo_params.verbose, cube_ref=ref_t,
                                     grid_args=dict(fwhm=algo_params.fwhm, source_xy=algo_params.source_xy,
                                                    full_output=algo_params.full_output, rot_options=rot_options))
                if grid:
                    # returns of the single-pass grid (pca_fullfr.py:744-755)
                    if algo_params.source_xy is None:
                        if algo_params.full_output:
                            return host(res[0], np.float64), res[1]
                        return host(res, np.float64)
                    cubeout, finalfr, table, _ = res
                    if algo_params.full_output:
                        return host(cubeout, np.float64), host(finalfr, np.float64), table
                    return host(finalfr, np.float64)
                allfr, desc, adi, frame = res
                if algo_params.full_output:
                    return host(frame, np.float64), host(allfr, np.float64), host(desc), host(adi, np.float64)
                return host(frame, np.float64)
            raise ValueError("`adimsdi` mode not recognized")
    cube_ref_t = None
    if algo_params.cube_ref is not None:
        cube_ref_t = B.to_device_f32(algo_params.cube_ref)

    fo = bool(algo_params.full_output)
    add = {"start_time": None, "full_output": fo}
    if algo_params.cube_sig is not None:
        if cube.ndim == 4:
            raise NotImplementedError("cube_sig with a 4-D cube is outside the accelerated path")
        if not _is_array(algo_params.cube_sig) or tuple(algo_params.cube_sig.shape) != tuple(cube.shape):
            raise TypeError("`cube_sig` must be an array with the shape of `cube`")
        add["cube_sig"] = B.to_device_f32(algo_params.cube_sig)

    if cube.ndim == 4:
        torch = B._torch()
        nch, nz, ny, nx = cube.shape
        ncomp = algo_params.ncomp
        if not isinstance(ncomp, list):
            ncomps = [ncomp] * nch
        elif len(ncomp) != nch:
            ncomps = [ncomp] * nch            # the list is a PCA grid applied to every channel (pca_fullfr.py:548-551)
        else:
            ncomps = ncomp
        fwhm = algo_params.fwhm
        fwhms = [fwhm] * nch if np.isscalar(fwhm) else fwhm
        # fast path: plain ADI, one integer ncomp for all channels, final frame only -> batched small stages
        mask_val = rot_options.get("mask_val", np.nan)
        mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
        same_k = all(isinstance(kc, (int, np.integer)) and kc == ncomps[0] for kc in ncomps)
        if (not fo and same_k and cube_ref_t is None and algo_params.source_xy is None and algo_params.batch is None
                and algo_params.mask_rdi is None and algo_params.smooth is None and not algo_params.left_eigv
                and _s(algo_params.imlib) == "vip-fft" and (mv_nan or mask_val == 0)
                and rot_options.get("edge_blend") in (None, "") and 0 < int(ncomps[0]) <= min(64, nz) and nz <= 512
                and _s(algo_params.collapse) in B.COLLAPSE_MODES and _s(algo_params.collapse) != "stim"
                and not (_s(algo_params.collapse) == "wmean" and algo_params.weights is None)):
            angles = check_pa_vector(np.asarray(algo_params.angle_list, dtype=np.float64))
            if angles.shape[0] != nz:
                raise ValueError("`angle_list` vector has wrong length. It must equal the number of frames in the cube")
            ifs = _adi_pca_channels_batched(cube_t, angles, int(ncomps[0]), _s(algo_params.scaling),
                                            algo_params.mask_center_px, _s(algo_params.collapse), algo_params.weights,
                                            mv_nan)
            frame = B.collapse(ifs, _s(algo_params.collapse_ifs))
            return host(frame, np.float64)
        outs = []
        # the channels are independent: issue them round-robin on two streams in asynchronous mode, so that the
        # latency-bound eigensolver of one channel runs beside the derotation of the previous one (unless the
        # caller already pipelines whole calls itself)
        pipelined = nch > 1 and not B.is_async()
        cur = torch.cuda.current_stream()
        streams = B.side_streams(2, cube_t.device.index) if pipelined else [cur]
        if pipelined:
            B.set_async(True)
        try:
            for ch in range(nch):
                st = streams[ch % len(streams)]
                if pipelined and ch < len(streams):
                    st.wait_stream(cur)
                with torch.cuda.stream(st):
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
            if pipelined:
                for st in streams:
                    cur.wait_stream(st)
                B.check_deferred()
        finally:
            if pipelined:
                for st in streams:              # (also on the error path, before the per-channel buffers are released)
                    cur.wait_stream(st)
                B.set_async(False)
        grid_ch = [isinstance(kc, (tuple, list)) for kc in ncomps]
        if any(grid_ch):
            # per-channel PCA grid (pca_fullfr.py:614-617,626-629,639-646): one collapsed frame per grid entry
            if not all(grid_ch):
                raise TypeError("`ncomp` must be a grid for every channel or for none")
            if algo_params.source_xy is not None:
                # per-channel grids scored by the S/N at source_xy (pca_fullfr.py:604-612): the channel's frame is its
                # best-S/N frame; returns (per-channel grid cubes, frame, tables[, ifs_adi_frames]).  `med_of_npcs`
                # takes the reference's median over axis 0 of the stacked grid cubes -- the channel axis (:718-719).
                cubes_ch = torch.stack([o[0] for o in outs])                   # (nch, n_grid, y, x)
                ifs = torch.stack([o[1] for o in outs])
                tables = [o[2] for o in outs]
                frame = B.collapse(ifs, _s(algo_params.collapse_ifs))
                if algo_params.med_of_npcs:
                    cubes_ch = B.collapse(cubes_ch.reshape(nch, -1, 1).contiguous(), "median").reshape(cubes_ch.shape[1:])
                if fo:
                    return host(cubes_ch, np.float64), host(frame, np.float64), tables, host(ifs, np.float64)
                return host(frame, np.float64)
            cubes_ch = torch.stack([o[0] if fo else o for o in outs])          # (nch, n_grid, y, x)
            final = torch.stack([B.collapse(cubes_ch[:, i].contiguous(), _s(algo_params.collapse_ifs))
                                 for i in range(cubes_ch.shape[1])])
            if algo_params.med_of_npcs:
                final = B.collapse(final, "median")
            if fo:
                return host(final, np.float64), [o[1] for o in outs], host(cubes_ch, np.float64)
            return host(final, np.float64)
        ifs = torch.stack([o[-1] if fo else o for o in outs])
        frame = B.collapse(ifs, _s(algo_params.collapse_ifs))
        if fo and algo_params.source_xy is not None:
            # frame rejection per channel (pca_fullfr.py:619-623,783-788): (frame, recon_cube, residuals, residuals_, ifs)
            recon = torch.stack([o[0] for o in outs])
            res = torch.stack([o[1] for o in outs])
            resd = torch.stack([o[2] for o in outs])
            return host(frame, np.float64), host(recon), host(res), host(resd), host(ifs, np.float64)
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
    if isinstance(algo_params.ncomp, (tuple, list)):
        if algo_params.source_xy is not None:
            # S/N-scored grid (pca_fullfr.py:706-713,779-783,789-790): (final_residuals_cube, frame, table) or the frame
            cubeout, finalfr, table, _opt = out
            if algo_params.med_of_npcs:
                cubeout = B.collapse(cubeout, "median")
            return (host(cubeout), host(finalfr), table) if fo else host(finalfr)
        # PCA grid (pca_fullfr.py:716-717,776-777,792-793): cube of final frames [, list of PCs]
        cubeout, pclist = out if fo else (out, None)
        if algo_params.med_of_npcs:
            cubeout = B.collapse(cubeout, "median")
        return (host(cubeout), pclist) if fo else host(cubeout)
    if algo_params.source_xy is not None:
        if fo:
            recon_cube, residuals_cube, residuals_cube_, frame = out
            return host(frame), host(recon_cube), host(residuals_cube), host(residuals_cube_)
        return host(out)
    if fo:
        pcs, recon, residuals_cube, residuals_cube_, frame = out
        if not dev_in:
            return tuple(B.to_host_many([frame, pcs, recon, residuals_cube, residuals_cube_], [out_dtype] * 5))
        return host(frame), host(pcs), host(recon), host(residuals_cube), host(residuals_cube_)
    return host(out)


def pca_many(cubes, angle_lists, depth=2, **kwargs):
    """``[pca(c, a, **kwargs) for c, a in zip(cubes, angle_lists)]`` for independent cubes (survey mode, the
    contrast-curve / NEGFC loops of the reference, which it parallelises over processes with ``nproc``), issued
    through ``depth`` streams in the library's asynchronous mode so that the eigensolver of one cube runs beside the
    derotation of the previous one.  ``full_output`` is not supported; numpy cubes are uploaded
    one ahead.  Returns a list of final frames (numpy in -> numpy out, cuda tensors in -> cuda tensors out)."""
    if kwargs.get("full_output"):
        raise NotImplementedError("pca_many returns final frames only")
    kwargs = dict(kwargs, full_output=False, verbose=False)
    torch = B.require_gpu()

    def upload(c):
        # a float64 cube stays float64 on the device: pca() then takes the route that carries the temporal mean in float64
        # (_float64_fused), exactly as pca(c, a) on the caller's array does -- rounding it to float32 here cost 2e-3 on the
        # frame of golden g28 (round-5 ADVICE)
        if B.is_device_tensor(c):
            return c.contiguous() if c.dtype == torch.float64 else B.to_device_f32(c)
        if isinstance(c, np.ndarray) and c.dtype == np.float64:
            return torch.from_numpy(np.ascontiguousarray(c)).to(torch.device("cuda", torch.cuda.current_device()))
        return B.to_device_f32(c)
    n_items = len(cubes)
    if n_items != len(angle_lists):
        raise ValueError("cubes and angle_lists must have the same length")
    depth = max(1, min(int(depth), n_items)) if n_items else 1
    streams = B.side_streams(depth)       # cached: every stream owns a context with its own workspaces (backend._ctx_cache)
    dev_in = [B.is_device_tensor(c) for c in cubes]
    outs = [None] * n_items
    # numpy cubes: uploaded by a thread of their own on a copy stream, up to two ahead of the cube whose kernels this thread is
    # enqueueing (the interpreter lock is released inside the blocking pageable copy): the link never waits for the ~0.4 ms of
    # host work per call (8.45 -> 8.1 ms per C2 cube, the upload alone being 8.05)
    feeder = None
    if n_items > 1 and not any(dev_in) and os.environ.get("VIPMI_HOSTIN", "1") != "0":
        import queue
        import threading
        dev = torch.cuda.current_device()
        copy_stream = B.side_streams(depth + 1, dev)[depth]
        copy_stream.wait_stream(torch.cuda.current_stream())
        q = queue.Queue(maxsize=2)

        def uploader():
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(copy_stream):
                    for c in cubes:
                        t_ = upload(c)
                        q.put((t_, copy_stream.record_event()))
            except BaseException as e:
                q.put(e)

        feeder = threading.Thread(target=uploader, name="vipmi-upload", daemon=True)
        feeder.start()
    B.set_async(True)
    try:
        cur = torch.cuda.current_stream()
        for i, (c, a) in enumerate(zip(cubes, angle_lists)):
            st = streams[i % depth]
            st.wait_stream(cur)
            if feeder is not None:
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                t, ev = item
                st.wait_event(ev)
                t.record_stream(st)
                with torch.cuda.stream(st):
                    outs[i] = pca(t, a, **kwargs)
                del t, item
                continue
            with torch.cuda.stream(st):
                t = upload(c)
                outs[i] = pca(t, a, **kwargs)
        for st in streams:
            st.synchronize()
        B.check_deferred()
    finally:
        if feeder is not None:
            while feeder.is_alive():            # (drain after an early exit: the uploader may be blocked on a full queue)
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            feeder.join()
        for st in streams:                      # (a no-op after the normal path; joins the streams on the error path)
            st.synchronize()
        B.set_async(False)
    res = []
    for i, o in enumerate(outs):
        if dev_in[i]:
            res.append(o)
        else:
            c = cubes[i]
            res.append(B.to_host(o, c.dtype if c.dtype.kind == "f" else np.float64))
    return res
