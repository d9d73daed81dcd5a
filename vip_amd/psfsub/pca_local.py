"""Annular PCA: drop-in for ``vip_hci.psfsub.pca_annular`` (reference psfsub/pca_local.py:39-70
PCA_ANNULAR_Params, :73-462 pca_annular, :594-827 _pca_adi_rdi, :830-909 do_pca_patch) for 3-D ADI
cubes.

The reference runs one SVD per (annulus segment, frame) on the PA-thresholded library rows of the
segment matrix.  Here each segment matrix is gathered once on the device, its Gram matrix is formed on
the matrix cores, and every frame's library PCA is obtained from the corresponding Gram sub-block
(SURVEY.md 8(a-ann)): residual_j = x_j - M_lib^T (E_k L_k^-1 E_k^T) G[lib, j], batched over frames.

``ncomp``: int, per-annulus tuple, or a list (several truncations of one decomposition -> 4-D ``cube_out`` /
``cube_der`` and a list of frames).  4-D cubes without ``scale_list`` run the same path per spectral channel.
``cube_ref`` (RDI: reference frames stacked on every library) and ``cube_sig`` are supported.  Not accelerated
(NotImplementedError): ``scale_list`` (mSDI), ``left_eigv``, ``ncomp='auto'``.
"""
import ctypes
from collections import OrderedDict
from dataclasses import dataclass
from enum import Enum
from typing import List, Tuple, Union

import os

import numpy as np

from .. import backend as B
from ..config.paramenum import ALGO_KEY, Collapse, Imlib, Interpolation, SvdMode
from ..config.utils_param import separate_kwargs_dict, setup_parameters
from ..preproc.derotation import _define_annuli, _find_indices_adi_all
from ..preproc.parangles import check_pa_vector
from ..var.shapes import get_annulus_segments

AUTO = "auto"


@dataclass
class PCA_ANNULAR_Params:
    """Parameters of ``pca_annular`` (field order == positional order of the reference)."""

    cube: np.ndarray = None
    angle_list: np.ndarray = None
    cube_ref: np.ndarray = None
    scale_list: np.ndarray = None
    radius_int: int = 0
    fwhm: float = 4
    asize: float = 4
    n_segments: Union[int, List[int], str] = 1
    delta_rot: Union[float, Tuple[float], List[float]] = (0.1, 1)
    delta_sep: Union[float, Tuple[float], List[float]] = (0.1, 1)
    ncomp: Union[int, Tuple, np.ndarray, str] = 1
    svd_mode: Enum = SvdMode.LAPACK
    nproc: int = 1
    min_frames_lib: int = 2
    max_frames_lib: int = 200
    tol: float = 1e-1
    scaling: Enum = None
    imlib: Enum = Imlib.VIPFFT
    interpolation: Enum = Interpolation.LANCZOS4
    collapse: Enum = Collapse.MEDIAN
    collapse_ifs: Enum = Collapse.MEAN
    ifs_collapse_range: Union[str, Tuple[int]] = "all"
    theta_init: int = 0
    weights: np.ndarray = None
    cube_sig: np.ndarray = None
    full_output: bool = False
    verbose: bool = True
    left_eigv: bool = False


def _s(x):
    return str(getattr(x, "value", x)) if x is not None else None


def annulus_plan(shape, angle_list, radius_int, fwhm, asize, n_segments, delta_rot, ncomp, min_frames_lib,
                 max_frames_lib, theta_init=0):
    """Host-side plan of ``_pca_adi_rdi`` (pca_local.py:628-707): for every annulus segment, the flat
    pixel indices, the number of PCs and the per-frame library index lists.  Pure index arithmetic
    (bit-exact with the reference); returns a list of dicts."""
    n = angle_list.shape[0]
    y, x = shape
    n_annuli = int((y / 2 - radius_int) / asize)
    if isinstance(delta_rot, tuple):
        delta_rot = np.linspace(delta_rot[0], delta_rot[1], num=n_annuli)
    elif np.isscalar(delta_rot):
        delta_rot = [delta_rot] * n_annuli
    else:
        if len(delta_rot) != n_annuli:
            raise TypeError("If delta_rot is a list it should have n_annuli elements.")
    if isinstance(n_segments, int):
        n_segments = [n_segments for _ in range(n_annuli)]
    elif n_segments == "auto":
        n_segments = [2, 3]
        ld = 2 * np.tan(360 / 4 / 2) * asize        # argument in radians, as in the reference (:648)
        for i in range(2, n_annuli):
            ang = np.rad2deg(2 * np.arctan(ld / (2 * i * asize)))
            n_segments.append(int(np.ceil(360 / ang)))
    plan = []
    lib_memo = {}                     # libraries by exclusion windows: shared by the annuli whose thresholds give the same windows
    for ann in range(n_annuli):
        if isinstance(ncomp, (tuple, np.ndarray)):
            if len(ncomp) != n_annuli:
                raise TypeError("If `ncomp` is a tuple, its length must match the number of annuli")
            k_ann = int(ncomp[ann])
        else:
            k_ann = int(ncomp)
        pa_thr, inner_radius, ann_center = _define_annuli(angle_list, ann, n_annuli, fwhm, radius_int, asize,
                                                          delta_rot[ann], n_segments[ann], False, True)
        segs = get_annulus_segments(np.zeros((y, x)), inner_radius, asize, n_segments[ann], theta_init)
        if pa_thr != 0:
            libs = _find_indices_adi_all(angle_list, pa_thr, truncate=True, max_frames=max_frames_lib, memo=lib_memo)
            for fr, li in enumerate(libs):
                if li.shape[0] < min_frames_lib:
                    msg = "Too few frames left in the PCA library. Accepted indices length ({:.0f}) less than {:.0f}. "
                    msg += "Try decreasing either delta_rot or min_frames_lib."
                    raise RuntimeError(msg.format(len(li), min_frames_lib))
        else:
            libs = [np.arange(n, dtype=np.int32) for _ in range(n)]
        for yy, xx in segs:
            plan.append(dict(ann=ann, pix=(yy.astype(np.int64) * x + xx).astype(np.int32), ncomp=k_ann,
                             libs=libs, pa_thr=pa_thr, inner_radius=inner_radius, ann_center=ann_center))
    return plan


# The plan is pure index arithmetic on (frame shape, angles, geometry parameters), but it costs tens of milliseconds
# of host time (n_annuli x n library selections) -- as much as the device work of a C3-sized call.  Callers such as
# contrast curves or fake-companion loops repeat the same geometry hundreds of times, so the last few plans are kept,
# together with the device copies of their pixel / library index arrays.
_PLAN_CACHE = OrderedDict()
_PLAN_CACHE_SIZE = 8


def _freeze(v):
    if isinstance(v, np.ndarray):
        return ("nd", v.shape, v.dtype.str, v.tobytes())
    if isinstance(v, (list, tuple)):
        return (type(v).__name__,) + tuple(_freeze(x) for x in v)
    return v


def cached_annulus_plan(shape, angle_list, radius_int, fwhm, asize, n_segments, delta_rot, ncomp, min_frames_lib,
                        max_frames_lib, theta_init=0):
    """``annulus_plan`` through a small LRU cache; returns (plan, dev) where ``dev`` is a dict the caller may use to
    keep device-side copies that belong to this plan."""
    key = _freeze((tuple(shape), np.ascontiguousarray(angle_list, dtype=np.float64), radius_int, fwhm, asize, n_segments,
                   delta_rot, ncomp, min_frames_lib, max_frames_lib, theta_init))
    hit = _PLAN_CACHE.get(key)
    if hit is None:
        hit = (annulus_plan(shape, angle_list, radius_int, fwhm, asize, n_segments, delta_rot, ncomp, min_frames_lib,
                            max_frames_lib, theta_init), {})
        _PLAN_CACHE[key] = hit
        while len(_PLAN_CACHE) > _PLAN_CACHE_SIZE:
            _PLAN_CACHE.popitem(last=False)
    else:
        _PLAN_CACHE.move_to_end(key)
    return hit


def _fused_front_plan(plan, n):
    """Host tables of the fused front (csrc/annular.hip annular_gram_all_f32 / annular_apply_all_f32): the pixel lists of all
    segments side by side, every segment padded to a whole number of K-slices of ``klen`` columns; ``pix_out`` is ``pix_all`` with
    -1 for the pixels that a LATER segment also holds (the reference applies the segments in order: the later one wins,
    pca_local.py:786-787); ``tile_seg``: the segment of every 128-column tile; ``seg_slice``: first slice of every segment."""
    sizes = [int(sg["pix"].size) for sg in plan]
    total = sum(sizes)
    klen = 512
    for cand in (2048, 1024, 512):                # the longest slice whose padding costs at most 8 % more columns
        if sum(-(-sz // cand) * cand for sz in sizes) <= 1.08 * total:
            klen = cand
            break
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + -(-sz // klen) * klen)
    Ptot = offs[-1]
    pix_all = np.full(Ptot, -1, dtype=np.int32)
    tile_seg = np.full(Ptot // 128, -1, dtype=np.int32)
    for si, sg in enumerate(plan):
        pix_all[offs[si]:offs[si] + sizes[si]] = sg["pix"]
        tile_seg[offs[si] // 128:(offs[si] + sizes[si] + 127) // 128] = si
    pix_out = pix_all.copy()
    live = np.nonzero(pix_all >= 0)[0]
    # last occurrence of every pixel wins: np.unique on the reversed list returns the first index there = the last one here
    rev = pix_all[live][::-1]
    _u, first_rev = np.unique(rev, return_index=True)
    keep = np.zeros(live.size, dtype=bool)
    keep[live.size - 1 - first_rev] = True
    pix_out[live[~keep]] = -1
    seg_slice = (np.asarray(offs, dtype=np.int64) // klen).astype(np.int32)
    return dict(klen=klen, Ptot=Ptot, pix_all=pix_all, pix_out=pix_out, tile_seg=tile_seg, seg_slice=seg_slice, offs=offs)


def _pack_libs(libs):
    n = len(libs)
    max_lib = max(len(li) for li in libs)
    idx = np.zeros((n, max_lib), dtype=np.int32)
    ln = np.zeros(n, dtype=np.int32)
    for j, li in enumerate(libs):
        idx[j, :len(li)] = li
        ln[j] = len(li)
    return idx, ln, max_lib


@B.with_rotation
def _pca_adi_rdi(cube, angle_list, radius_int=0, fwhm=4, asize=2, n_segments=1, delta_rot=1, ncomp=1,
                 svd_mode="lapack", nproc=None, min_frames_lib=2, max_frames_lib=200, tol=1e-1,
                 scaling=None, imlib="vip-fft", interpolation="lanczos4", collapse="median",
                 full_output=False, verbose=1, cube_ref=None, theta_init=0, weights=None, cube_sig=None,
                 left_eigv=False, cube64=None, **rot_options):
    """Device version of the reference's ``_pca_adi_rdi``; ``cube`` is a float32 cuda tensor.  ``cube64``: the same cube as a
    float64 cuda tensor (the caller's dtype): every segment matrix is then centred in float64 first (csrc/pca_f64.hip) and
    the float32 kernels work on what is left -- the reference's do_pca_patch keeps float64 (pca_local.py:830-909)."""
    torch = B._torch()
    if cube.ndim != 3:
        raise TypeError("Input array is not a cube or 3d array")
    if cube.shape[0] != np.asarray(angle_list).shape[0]:
        raise TypeError("Input vector or parallactic angles has wrong length")
    if left_eigv:
        raise NotImplementedError("left_eigv is outside the accelerated annular path")
    if cube_ref is not None and (cube_ref.ndim != 3 or tuple(cube_ref.shape[1:]) != tuple(cube.shape[1:])):
        raise TypeError("`cube_ref` must be a cube with the frame size of `cube`")
    if cube_sig is not None and tuple(cube_sig.shape) != tuple(cube.shape):
        raise TypeError("`cube_sig` must have the shape of `cube`")
    nref = 0 if cube_ref is None else int(cube_ref.shape[0])
    if isinstance(ncomp, str):
        raise NotImplementedError("ncomp='auto' is outside the accelerated annular path")
    ks = None
    if isinstance(ncomp, list):          # several truncations of one decomposition (pca_local.py:665-668,892-902)
        ks = np.asarray([int(k) for k in ncomp], dtype=np.int32)
        if ks.size == 0 or ks.min() <= 0:
            raise ValueError("every ncomp of the list must be a positive integer")
    B.check_imlib(imlib, interpolation)         # 'vip-fft' or 'opencv'; the decorator selects the rotation
    if _s(collapse) is not None and (_s(collapse) not in B.COLLAPSE_MODES or _s(collapse) == "stim"):
        raise TypeError("mode not recognized")                      # cube_collapse, subsampling.py:113-114
    n, y, x = cube.shape
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=np.float64))
    plan, plan_dev = cached_annulus_plan((y, x), angle_list, radius_int, fwhm, asize, n_segments, delta_rot,
                                         ncomp if ks is None else int(ks.max()),
                                         min_frames_lib if nref == 0 else 0,      # with a reference cube any library size
                                         max_frames_lib, theta_init)             # is accepted (pca_local.py:868-870)
    if verbose:
        print("N annuli = {}, FWHM = {:.3f}".format(int((y / 2 - radius_int) / asize), fwhm))
    dev = cube.device.index
    P = y * x
    cube_out = torch.zeros_like(cube) if ks is None else torch.zeros((len(ks),) + tuple(cube.shape), dtype=cube.dtype,
                                                                     device=cube.device)
    scaling = _s(scaling)
    pad_ok = scaling not in ("spat-mean", "spat-standard")
    lib_cache = plan_dev.setdefault(("libs", dev), {})
    pix_cache = plan_dev.setdefault(("pix", dev, pad_ok), {})

    def pix_of(si, seg):
        pix = pix_cache.get(si)
        if pix is None:
            pix_h = seg["pix"]
            # pad to a multiple of 4 columns (-1 = zero column) so rows are 16-byte aligned; zero columns change
            # neither the Gram matrix nor temporal statistics (spatial scaling needs the exact row length)
            if pix_h.size % 4 and pad_ok:
                pix_h = np.concatenate([pix_h, np.full(4 - pix_h.size % 4, -1, dtype=np.int32)])
            pix = pix_cache[si] = torch.from_numpy(pix_h).to(cube.device)
        return pix

    def libs_of(seg):
        key = (id(seg["libs"]), nref)
        if key not in lib_cache:
            libs = seg["libs"]
            if nref:
                head = np.arange(nref, dtype=np.int64)
                libs = [np.array([r]) for r in range(nref)] + [np.concatenate((head, np.asarray(li, dtype=np.int64) + nref))
                                                               for li in libs]
            idx, ln, max_lib = _pack_libs(libs)
            lib_cache[key] = (torch.from_numpy(idx).to(cube.device), torch.from_numpy(ln).to(cube.device), max_lib)
        return lib_cache[key]

    def seg_matrix(si, seg):
        """Gathered / scaled segment matrix (+ cube_sig part, reference rows): everything before the decomposition."""
        ctx = B.get_context(dev)                       # (one context per stream)
        pix = pix_of(si, seg)
        npx = int(pix.numel())
        A = B.empty((n, npx), device=dev)
        ctx.call("vipmi_gather_f32", B.ptr(cube), n, P, B.ptr(pix), npx, B.ptr(A))
        if scaling is not None:
            A = B.scale(A, scaling)
        S = None
        if cube_sig is not None:
            S = B.empty((n, npx), device=dev)
            ctx.call("vipmi_gather_f32", B.ptr(cube_sig), n, P, B.ptr(pix), npx, B.ptr(S))
            A = B.lincomb(A, S, 1.0, -1.0)
        if nref:
            Aref = B.empty((nref, npx), device=dev)
            ctx.call("vipmi_gather_f32", B.ptr(cube_ref), nref, P, B.ptr(pix), npx, B.ptr(Aref))
            if scaling is not None:
                Aref = B.scale(Aref, scaling)
            A = torch.cat((Aref, A))
        return A, S, pix, npx

    def seg_scatter(R, S, pix, npx, dst):
        ctx = B.get_context(dev)
        R = R[nref:]
        if S is not None:
            R = B.lincomb(R, S, 1.0, 1.0)
        R = R.contiguous()
        ctx.call("vipmi_scatter_f32", B.ptr(R), n, P, B.ptr(pix), npx, B.ptr(dst))

    def do_segment(si, seg):
        ctx = B.get_context(dev)                       # (one context per stream)
        pix = pix_of(si, seg)
        npx = int(pix.numel())
        A = B.empty((n, npx), device=dev)
        ctx.call("vipmi_gather_f32", B.ptr(cube), n, P, B.ptr(pix), npx, B.ptr(A))
        if scaling is not None:
            A = B.scale(A, scaling)
        # cube_sig (pca_local.py:721-724,862-866,887-891): libraries and projections from A - S (S unscaled), model
        # subtracted from A:  A - proj(A - S) = (A_emp - proj(A_emp)) + S
        S = None
        if cube_sig is not None:
            S = B.empty((n, npx), device=dev)
            ctx.call("vipmi_gather_f32", B.ptr(cube_sig), n, P, B.ptr(pix), npx, B.ptr(S))
            A = B.lincomb(A, S, 1.0, -1.0)
        # cube_ref (pca_local.py:716-720,879-885): the reference frames of the segment, scaled on their own, are stacked
        # on top of every frame's library -> one matrix of nref + n rows whose first nref rows only serve as library
        nrow = n + nref
        if nref:
            Aref = B.empty((nref, npx), device=dev)
            ctx.call("vipmi_gather_f32", B.ptr(cube_ref), nref, P, B.ptr(pix), npx, B.ptr(Aref))
            if scaling is not None:
                Aref = B.scale(Aref, scaling)
            A = torch.cat((Aref, A))
        idx_t, ln_t, max_lib = libs_of(seg)
        if ks is None:
            R = B.empty((nrow, npx), device=dev)
            ctx.call("vipmi_annular_residuals_f32", B.ptr(A), nrow, npx, B.ptr(idx_t), B.ptr(ln_t), max_lib,
                     int(seg["ncomp"]), B.ptr(R))
            R = R[nref:]
            if S is not None:
                R = B.lincomb(R, S, 1.0, 1.0)
            R = R.contiguous()
            ctx.call("vipmi_scatter_f32", B.ptr(R), n, P, B.ptr(pix), npx, B.ptr(cube_out))
        else:
            R = B.empty((len(ks), nrow, npx), device=dev)
            ctx.call("vipmi_annular_residuals_multi_f32", B.ptr(A), nrow, npx, B.ptr(idx_t), B.ptr(ln_t), max_lib,
                     ks.ctypes.data_as(ctypes.c_void_p), len(ks), B.ptr(R))
            for nn in range(len(ks)):
                Rn = R[nn, nref:]
                if S is not None:
                    Rn = B.lincomb(Rn, S, 1.0, 1.0)
                Rn = Rn.contiguous()
                ctx.call("vipmi_scatter_f32", B.ptr(Rn), n, P, B.ptr(pix), npx, B.ptr(cube_out[nn]))

    def do_segment_f64(si, seg):
        """The float64 route of one segment: A = D + 1 mu^T with the per-pixel temporal mean mu in float64,
        G = D D^T (+ the offset terms when the matrix is not scaled), libraries gathered from G by the solver,
        residuals = (I - C) D + rho mu^T."""
        ctx = B.get_context(dev)
        pix = pix_of(si, seg)
        npx = int(pix.numel())
        A64 = cube64.reshape(n, P).index_select(1, pix.clamp(min=0).long())
        npad = npx - int(seg["pix"].size)                           # (known on the host: no device read-back per segment)
        if npad:
            A64[:, npx - npad:] = 0.0                               # (the zero columns that pad the rows to 16 bytes)
        aug = B.empty((n + 1, npx), device=dev)                     # rows 0 .. n-1: D, row n: float32(mu)
        mu = torch.empty((npx,), dtype=torch.float64, device=cube.device)
        spat = scaling in ("spat-mean", "spat-standard")
        u = None
        if spat:
            # matrix_scaling(axis=1) of the segment matrix (var/shapes.py:740-781): diag(u) (A - m 1^T) = D + u mu^T, the frames'
            # means m and inverse standard deviations u over the segment's own pixels (the padding stays out)
            u = torch.empty((n,), dtype=torch.float64, device=cube.device)
            ctx.call("vipmi_spat_center_f64", B.ptr(A64), n, npx, npx - npad, 1 if scaling == "spat-standard" else 0, B.ptr(aug),
                     B.ptr(mu), B.ptr(aug[n]), B.ptr(u))
        else:
            mode = {None: 0, "temp-mean": 1, "temp-standard": 2}[scaling]
            ctx.call("vipmi_center_f64", B.ptr(A64), n, npx, mode, B.ptr(aug), B.ptr(mu), B.ptr(aug[n]))
        G = torch.empty((n, n), dtype=torch.float64, device=cube.device)
        ctx.call("vipmi_gram_f32", B.ptr(aug), n, npx, npx, B.ptr(G))
        if spat:
            ctx.call("vipmi_gram_offset_u_f64", B.ptr(aug), B.ptr(mu), B.ptr(u), n, npx, B.ptr(G))
        elif scaling is None:
            ctx.call("vipmi_gram_offset_f64", B.ptr(aug), B.ptr(mu), n, npx, B.ptr(G))
        idx_t, ln_t, max_lib = libs_of(seg)
        kseg = min(int(seg["ncomp"]), max_lib)
        work = torch.empty((n, max_lib, max_lib), dtype=torch.float64, device=cube.device)
        ev = torch.zeros((n, max_lib), dtype=torch.float64, device=cube.device)
        ec = torch.zeros((n, max_lib, max_lib), dtype=torch.float64, device=cube.device)
        ctx.call("vipmi_annular_eigh_f64", B.ptr(G), 1, n, B.ptr(idx_t), B.ptr(ln_t), max_lib, kseg, B.ptr(work), B.ptr(ev), B.ptr(ec))
        kk_ = np.array([int(seg["ncomp"])], dtype=np.int32)
        R = B.empty((1, n, npx), device=dev)
        if spat:
            ctx.call("vipmi_annular_apply_mu_u_f32", B.ptr(aug), n, npx, B.ptr(idx_t), B.ptr(ln_t), max_lib, max_lib, B.ptr(G), B.ptr(ev),
                     B.ptr(ec), kk_.ctypes.data_as(ctypes.c_void_p), 1, B.ptr(aug[n]), B.ptr(u), B.ptr(R))
        else:
            ctx.call("vipmi_annular_apply_mu_f32", B.ptr(aug), n, npx, B.ptr(idx_t), B.ptr(ln_t), max_lib, max_lib, B.ptr(G), B.ptr(ev),
                     B.ptr(ec), kk_.ctypes.data_as(ctypes.c_void_p), 1, B.ptr(aug[n]) if scaling is None else None, B.ptr(R))
        ctx.call("vipmi_scatter_f32", B.ptr(R[0].contiguous()), n, P, B.ptr(pix), npx, B.ptr(cube_out))

    f64_route = cube64 is not None and nref == 0 and cube_sig is None and ks is None and scaling in B.SCALE_MODES

    # Independent segments: issue the annuli round-robin on a few streams in asynchronous mode, so that the 400 per-frame
    # eigenproblems of one annulus (1.6 rounds of workgroups on 256 CUs) fill the idle tail of the previous one, and
    # one annulus' Gram / projection runs beside the other's eigensolver.  The last two annuli overlap by one pixel ring
    # and the later one must win (pca_local.py:786-787): they share a stream, which keeps their order.
    n_ann = plan[-1]["ann"] + 1 if plan else 0
    pipelined = n_ann >= 3 and not B.is_async() and not f64_route
    # Fused fronts (round 6): ONE gather of all segments, ONE ragged Gram product on the int8 matrix cores, ONE batched eigensolve,
    # ONE coefficient launch and ONE residual product that writes through the pixel list into cube_out -- for the plain ADI call
    # (no reference cube, no cube_sig, one ncomp per annulus, temporal or no scaling); VIPMI_ANNULAR_FUSED=0 disables it (the parity
    # tests compare the two routes)
    fused_env = os.environ.get("VIPMI_ANNULAR_FUSED", "")
    if plan and nref == 0 and cube_sig is None and pad_ok and fused_env != "0" and not (ks is not None and f64_route):
        m_all = max(max(len(li) for li in sg["libs"]) for sg in plan)
        k_all = min(m_all, max(int(sg["ncomp"]) for sg in plan) if ks is None else int(ks.max()))
        # (every size: measured from 64 x 101 x 101 to 400 x 512 x 512, 4- and 16-px annuli, the fused fronts are 0-45 % faster than
        #  the per-segment launches on two streams, tools/fused_threshold.py; the first version switched at 128 frames and 2^25 samples)
        if m_all <= 512 and k_all <= 64 and n * len(plan) * m_all * m_all * 16 <= 8e9 and P < 2 ** 31:
            fp_ = plan_dev.get(("fused", dev))
            if fp_ is None:
                h = _fused_front_plan(plan, n)
                total = n * len(plan)
                idx_h = np.zeros((total, m_all), dtype=np.int32)
                len_h = np.zeros(total, dtype=np.int32)
                for si, sg in enumerate(plan):
                    ih, lh, ml = _pack_libs(sg["libs"])
                    idx_h[si * n:(si + 1) * n, :ml] = ih
                    len_h[si * n:(si + 1) * n] = lh
                kseg_h = np.asarray([min(int(sg["ncomp"]), int(sg["pix"].size)) for sg in plan], dtype=np.int32)
                up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cube.device)
                fp_ = plan_dev[("fused", dev)] = dict(klen=h["klen"], Ptot=h["Ptot"], pix_all=up(h["pix_all"]), pix_out=up(h["pix_out"]),
                                                      tile_seg=up(h["tile_seg"]), seg_slice=up(h["seg_slice"]), idx=up(idx_h),
                                                      len=up(len_h), kseg=up(kseg_h), m_all=m_all, k_all=k_all)
            ctx = B.get_context(dev)
            nseg, Ptot = len(plan), fp_["Ptot"]
            A_all = B.empty((n, Ptot), device=dev)
            G_all = torch.empty((nseg, n, n), dtype=torch.float64, device=cube.device)
            mu32 = None
            if f64_route:
                # a float64 cube: gather + centring in float64 in one pass (A_all = D = float32(cube - 1 mu^T) (/ sd)), the Gram
                # matrices of D -- plus the float64 offset terms when the matrix is not scaled --, and residuals = (I - C) D + rho mu^T
                mu64 = torch.empty((Ptot,), dtype=torch.float64, device=cube.device)
                mu32_t = B.empty((Ptot,), device=dev)
                mode = {None: 0, "temp-mean": 1, "temp-standard": 2}[scaling]
                ctx.call("vipmi_annular_gram_all_f64", B.ptr(cube64), n, P, B.ptr(fp_["pix_all"]), Ptot, fp_["klen"],
                         B.ptr(fp_["seg_slice"]), nseg, mode, B.ptr(A_all), B.ptr(mu64), B.ptr(mu32_t), B.ptr(G_all))
                mu32 = mu32_t if scaling is None else None
            elif scaling is None:
                ctx.call("vipmi_annular_gram_all_f32", B.ptr(cube), n, P, B.ptr(fp_["pix_all"]), Ptot, fp_["klen"],
                         B.ptr(fp_["seg_slice"]), nseg, B.ptr(A_all), B.ptr(G_all))
            else:
                # temporal scalings are per pixel column: the whole gathered matrix at once, then the same ragged Gram
                ctx.call("vipmi_gather_f32", B.ptr(cube), n, P, B.ptr(fp_["pix_all"]), Ptot, B.ptr(A_all))
                A_all = B.scale(A_all, scaling)
                ctx.call("vipmi_annular_gram_all_f32", None, n, P, None, Ptot, fp_["klen"], B.ptr(fp_["seg_slice"]), nseg,
                         B.ptr(A_all), B.ptr(G_all))          # (cube NULL: A_all already holds the matrix)
            m_all, k_all = fp_["m_all"], fp_["k_all"]
            H_all = torch.empty((nseg * n, m_all, m_all), dtype=torch.float64, device=cube.device)
            ev_all = torch.empty((nseg * n, m_all), dtype=torch.float64, device=cube.device)
            ec_all = torch.empty((nseg * n, m_all, m_all), dtype=torch.float64, device=cube.device)
            ctx.call("vipmi_annular_eigh_f64", B.ptr(G_all), nseg, n, B.ptr(fp_["idx"]), B.ptr(fp_["len"]), m_all, k_all,
                     B.ptr(H_all), B.ptr(ev_all), B.ptr(ec_all))
            if ks is None:
                ctx.call("vipmi_annular_apply_all_f32", B.ptr(A_all), n, Ptot, B.ptr(fp_["tile_seg"]), B.ptr(fp_["pix_out"]), nseg,
                         B.ptr(fp_["idx"]), B.ptr(fp_["len"]), m_all, B.ptr(G_all), B.ptr(ev_all), B.ptr(ec_all), B.ptr(fp_["kseg"]),
                         k_all, P, B.ptr(cube_out), B.ptr(mu32))
            else:
                # a list of ncomp (pca_local.py:665-668,892-902): one decomposition with max(ncomp), the coefficient / residual stage
                # once per truncation into its own residual cube
                npx_h = np.asarray([int(sg["pix"].size) for sg in plan], dtype=np.int32)
                ksegs = [torch.from_numpy(np.minimum(npx_h, int(kk_))).to(cube.device) for kk_ in ks]       # (alive until the sync below)
                for nn in range(len(ks)):
                    ctx.call("vipmi_annular_apply_all_f32", B.ptr(A_all), n, Ptot, B.ptr(fp_["tile_seg"]), B.ptr(fp_["pix_out"]), nseg,
                             B.ptr(fp_["idx"]), B.ptr(fp_["len"]), m_all, B.ptr(G_all), B.ptr(ev_all), B.ptr(ec_all),
                             B.ptr(ksegs[nn]), k_all, P, B.ptr(cube_out[nn]), None)
                plan_dev[("fused_ksegs", dev)] = ksegs            # (kept with the plan: the launches above may still be queued)
            plan = []                                 # (nothing left for the per-segment routes below)
            pipelined = False
            f64_route = False
    if f64_route:
        for si, seg in enumerate(plan):
            do_segment_f64(si, seg)
    elif pipelined:
        for si, seg in enumerate(plan):               # index uploads before the fork
            pix_of(si, seg)
            libs_of(seg)
        cur = torch.cuda.current_stream()
        depth = max(2, min(int(os.environ.get("VIPMI_ANNULAR_STREAMS", "2")), n_ann - 1))
        streams = B.side_streams(depth, dev)
        lane_of = lambda a: (n_ann - 2) % depth if a == n_ann - 1 else a % depth
        # Staged variant: the libraries of ALL segments in ONE batched eigensolve between the per-segment Gram / sub-Gram
        # stage and the per-segment coefficient / residual stage.  With thousands of problems in one launch the solver runs
        # its latency-bound second half four problems per CU beside the register-resident tridiagonalisations of the
        # next ones (csrc/eigh_tri.hip); per-segment launches of ~400 problems cannot do that.  C3: 18.5 -> see NOTES.
        nrow = n + nref
        total = nrow * len(plan)
        m_all = max(libs_of(seg)[2] for seg in plan)
        k_all = min(m_all, max(int(seg["ncomp"]) for seg in plan) if ks is None else int(ks.max()))
        staged = (os.environ.get("VIPMI_ANNULAR_STAGED", "1") != "0" and total >= 512 and m_all <= 512 and k_all <= 64
                  and total * m_all * m_all * 16 <= 8e9)
        ctx0 = B.get_context(dev)
        if staged:
            H_all = torch.empty((total, m_all, m_all), dtype=torch.float64, device=cube.device)
            ev_all = torch.empty((total, m_all), dtype=torch.float64, device=cube.device)
            ec_all = torch.empty((total, m_all, m_all), dtype=torch.float64, device=cube.device)
            nact_all = torch.cat([libs_of(seg)[1] for seg in plan]).contiguous()
            # One eigensolve that gathers every library's sub-Gram matrix itself from its segment's Gram matrix
            # (vipmi_annular_eigh_f64): the index lists of all segments padded to m_all columns, the Gram matrices contiguous
            idx_all = torch.zeros((total, m_all), dtype=torch.int32, device=cube.device)
            for si, seg in enumerate(plan):
                it, _lt, ml = libs_of(seg)
                idx_all[si * nrow:(si + 1) * nrow, :ml] = it.reshape(nrow, ml)
            G_all = torch.empty((len(plan), nrow, nrow), dtype=torch.float64, device=cube.device)
        for st in streams:
            st.wait_stream(cur)
        B.set_async(True)
        try:
            if not staged:
                for si, seg in enumerate(plan):
                    with torch.cuda.stream(streams[lane_of(seg["ann"])]):
                        do_segment(si, seg)
            else:
                states = []
                for si, seg in enumerate(plan):
                    with torch.cuda.stream(streams[lane_of(seg["ann"])]):
                        A, S, pix, npx = seg_matrix(si, seg)
                        G = G_all[si]
                        B.get_context(dev).call("vipmi_gram_f32", B.ptr(A), nrow, npx, npx, B.ptr(G))
                        states.append((A, S, pix, npx, G))
                for st in streams:
                    cur.wait_stream(st)
                ctx0.call("vipmi_annular_eigh_f64", B.ptr(G_all), len(plan), nrow, B.ptr(idx_all), B.ptr(nact_all), m_all, k_all,
                          B.ptr(H_all), B.ptr(ev_all), B.ptr(ec_all))
                for st in streams:
                    st.wait_stream(cur)
                for si, seg in enumerate(plan):
                    with torch.cuda.stream(streams[lane_of(seg["ann"])]):
                        A, S, pix, npx, G = states[si]
                        idx_t, ln_t, max_lib = libs_of(seg)
                        kk_ = np.array([int(seg["ncomp"])], dtype=np.int32) if ks is None else ks
                        R = B.empty((len(kk_), nrow, npx), device=dev)
                        B.get_context(dev).call("vipmi_annular_apply_f32", B.ptr(A), nrow, npx, B.ptr(idx_t), B.ptr(ln_t),
                                                max_lib, m_all, B.ptr(G), B.ptr(ev_all[si * nrow:(si + 1) * nrow]),
                                                B.ptr(ec_all[si * nrow:(si + 1) * nrow]),
                                                kk_.ctypes.data_as(ctypes.c_void_p), len(kk_), B.ptr(R))
                        for nn in range(len(kk_)):
                            seg_scatter(R[nn], S, pix, npx, cube_out if ks is None else cube_out[nn])
            for st in streams:
                cur.wait_stream(st)
            B.check_deferred()
        finally:
            for st in streams:                  # (also on the error path: nothing the side streams still use may be
                cur.wait_stream(st)             # released before they are joined back)
            B.set_async(False)
    else:
        for si, seg in enumerate(plan):
            do_segment(si, seg)
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    if ks is not None:
        cube_der = torch.stack([B.derotate(cube_out[nn], angle_list, mask_nan=mv_nan, mask_zero=not mv_nan)
                                for nn in range(len(ks))])
        frames = list(B.collapse_batched(cube_der, _s(collapse), w=weights))
        if verbose:
            print("Done derotating and combining.")
        if full_output:
            return cube_out, cube_der, frames
        return frames
    cube_der = B.derotate(cube_out, angle_list, mask_nan=mv_nan, mask_zero=not mv_nan)
    frame = B.collapse(cube_der, _s(collapse), w=weights)
    if verbose:
        print("Done derotating and combining.")
    if full_output:
        return cube_out, cube_der, frame
    return frame


def pca_annular(*all_args: List, **all_kwargs: dict):
    """Annular ADI PCA on the MI355X.  Returns ``frame`` or ``(cube_out, cube_der, frame)``."""
    class_params, rot_options = separate_kwargs_dict(all_kwargs, PCA_ANNULAR_Params)
    algo_params = None
    if ALGO_KEY in rot_options.keys():
        algo_params = rot_options[ALGO_KEY]
        del rot_options[ALGO_KEY]
    if algo_params is None:
        algo_params = PCA_ANNULAR_Params(*all_args, **class_params)
    # by default, interpolate masked area before derotation if a mask is used (pca_local.py:242-245)
    if algo_params.radius_int and len(rot_options) == 0:
        rot_options["mask_val"] = 0
        rot_options["ker"] = 1
        rot_options["interp_zeros"] = True
    if algo_params.left_eigv:
        raise NotImplementedError("left_eigv is outside the accelerated path")
    cube = algo_params.cube
    if not (isinstance(cube, np.ndarray) or B.is_device_tensor(cube)):
        raise TypeError("`cube` must be a numpy ndarray")
    if algo_params.scale_list is not None:
        raise NotImplementedError("ADI+mSDI annular PCA (scale_list) is not accelerated yet (SURVEY 8(f))")
    if cube.ndim not in (3, 4):
        raise TypeError("Input array is not a cube or 3d array")
    dev_in = B.is_device_tensor(cube)
    out_dtype = None if dev_in else (cube.dtype if cube.dtype.kind == "f" else np.float64)

    def host(t):
        return t if dev_in else B.to_host(t, out_dtype)

    def host4(t):          # the reference builds the per-channel frames in a float64 array (pca_local.py:281)
        return t if dev_in else B.to_host(t, np.float64)

    # a float64 numpy cube is uploaded ONCE (838 MB at C2 size: 16 ms over PCIe): its float32 copy is made on the device, and the
    # float64 tensor itself serves the float64 route below (round 5 uploaded it twice: 44.8 ms per call where 29 suffice)
    cube64_t = None
    if cube.ndim == 3 and not dev_in and cube.dtype == np.float64:
        torch_ = B.require_gpu()
        cube64_t = torch_.from_numpy(np.ascontiguousarray(cube)).to(torch_.device("cuda", torch_.cuda.current_device()))
        cube_t = cube64_t.to(torch_.float32)
    else:
        cube_t = B.to_device_f32(cube)
    if cube.ndim == 4:
        # 4-D cube without scale_list: one annular ADI PCA per spectral channel, then collapse_ifs
        # (pca_local.py:279-325); ncomp / fwhm broadcast per channel
        torch = B._torch()
        nch = cube.shape[0]
        ncomp = algo_params.ncomp
        if not isinstance(ncomp, list) or len(ncomp) != nch:
            ncomp = [ncomp] * nch
        fwhm = algo_params.fwhm
        if np.isscalar(fwhm):
            fwhm = [fwhm] * nch
        if algo_params.cube_sig is not None:
            raise NotImplementedError("cube_sig with a 4-D cube is outside the accelerated annular path")
        ref_t = None
        if algo_params.cube_ref is not None:
            ref_t = B.to_device_f32(algo_params.cube_ref)
            if ref_t.ndim != 4 or ref_t.shape[0] != nch:
                raise TypeError("Ref cube has wrong format for 4d input cube")
        outs = []
        # independent channels: two streams in asynchronous mode (see the 4-D loop of psfsub/pca_fullfr.py)
        pipelined = nch > 1 and not B.is_async()
        cur = torch.cuda.current_stream()
        streams = B.side_streams(2, cube_t.device.index) if pipelined else [cur]
        if pipelined:
            B.set_async(True)
        try:
            for ch in range(nch):
                if isinstance(ncomp[ch], list):
                    raise NotImplementedError("a list of ncomp per channel is outside the accelerated annular path")
                st = streams[ch % len(streams)]
                if pipelined and ch < len(streams):
                    st.wait_stream(cur)
                with torch.cuda.stream(st):
                    fp = setup_parameters(algo_params, _pca_adi_rdi, cube=cube_t[ch], fwhm=fwhm[ch], ncomp=ncomp[ch],
                                          full_output=True, cube_ref=None if ref_t is None else ref_t[ch])
                    outs.append(_pca_adi_rdi(**fp, **rot_options))
            if pipelined:
                for st in streams:
                    cur.wait_stream(st)
                B.check_deferred()
        finally:
            if pipelined:
                for st in streams:              # (also on the error path, before the per-channel buffers are released)
                    cur.wait_stream(st)
                B.set_async(False)
        ifs = torch.stack([o[2] for o in outs])
        frame = B.collapse(ifs, _s(algo_params.collapse_ifs)) if algo_params.collapse_ifs is not None else ifs
        if algo_params.full_output:
            return (host(torch.stack([o[0] for o in outs])), host(torch.stack([o[1] for o in outs])),
                    host4(frame))
        return host4(frame)
    extra = {}
    if algo_params.cube_ref is not None:
        extra["cube_ref"] = B.to_device_f32(algo_params.cube_ref)
    if algo_params.cube_sig is not None:
        extra["cube_sig"] = B.to_device_f32(algo_params.cube_sig)
    # a float64 cube keeps its dtype through the decomposition (the reference's do_pca_patch): the float64 copy goes along and the
    # segment matrices are centred in float64 (plain ADI, every scaling; _pca_adi_rdi decides)
    if algo_params.cube_ref is None and algo_params.cube_sig is None:
        torch = B._torch()
        if dev_in and cube.dtype == torch.float64:
            extra["cube64"] = cube.to(cube_t.device).contiguous()
        elif not dev_in and cube.dtype == np.float64:
            extra["cube64"] = cube64_t
    fp = setup_parameters(algo_params, _pca_adi_rdi, cube=cube_t, full_output=True, **extra)
    cube_out, cube_der, frame = _pca_adi_rdi(**fp, **rot_options)
    if isinstance(frame, list):
        # list ncomp: the reference allocates cube_out / cube_der with np.zeros (float64, pca_local.py:666-668,800)
        frames = [host4(f) for f in frame]
        if algo_params.full_output:
            return host4(cube_out), host4(cube_der), frames
        return frames
    if algo_params.full_output:
        return host(cube_out), host(cube_der), host(frame)
    return host(frame)
