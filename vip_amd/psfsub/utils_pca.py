"""``pca_grid`` and ``pca_annulus`` of the reference's psfsub/utils_pca.py on the MI355X.

* ``pca_grid`` (reference :25-428): ONE decomposition with the largest number of PCs, then for every entry of the grid
  truncate -> subtract -> derotate -> collapse on the device; with ``source_xy`` every final frame is scored on the host
  by the S/N at that position (``vip_amd.metrics.snr``: a few dozen small aperture sums per frame) and the best number
  of PCs is returned with the table of S/Ns and fluxes.  What contrast-curve / NEGFC callers hammer.
* ``pca_annulus`` (reference :617-755): PCA-ADI / PCA-RDI restricted to one annulus (the NEGFC workhorse).

numpy in -> numpy out, cuda tensors in -> cuda tensors out.
"""
import numpy as np

from .. import backend as B
from ..metrics.snr_source import disk_pixels, frame_report, snr
from ..preproc.parangles import check_pa_vector
from ..var.coords import dist, frame_center
from ..var.shapes import center_mask_u8, get_annulus_segments


def _s(x):
    return str(getattr(x, "value", x)) if x is not None else None


def _prep_fullfr(cube_t, scaling, mask_center_px):
    """prepare_matrix(mode='fullfr') on the device (var/shapes.py:857-873)."""
    n, y, x = cube_t.shape
    m = cube_t.reshape(n, -1)
    if mask_center_px:
        mask = B.to_device_f32(center_mask_u8((y, x), mask_center_px).astype(np.float32)).to(B._torch().uint8)
        m = B.apply_mask(m, mask.reshape(-1), 0.0)
    if scaling is not None:
        m = B.scale(m, scaling)
    return m


def _annulus_pixels(shape, inrad, outrad):
    """flat pixel indices of prepare_matrix(mode='annular') (var/shapes.py:838-848): the one-segment annulus."""
    yy, xx = get_annulus_segments(shape, inrad, int(np.round(outrad - inrad)), nsegm=1)[0]
    return (yy.astype(np.int64) * shape[1] + xx).astype(np.int32)


def _gather_annulus(cube_t, pix_t, scaling):
    n = cube_t.shape[0]
    P = cube_t[0].numel()
    npx = int(pix_t.numel())
    ctx = B.get_context(cube_t.device.index)
    A = B.empty((n, npx), device=cube_t.device.index)
    ctx.call("vipmi_gather_f32", B.ptr(cube_t), n, P, B.ptr(pix_t), npx, B.ptr(A))
    if scaling is not None:
        A = B.scale(A, scaling)
    return A


def _get_snr(frame, y, x, fwhm, fmerit, exclude_negative_lobes):
    """utils_pca.py:239-277: S/N and flux figure of merit at (x, y)."""
    def one(y_, x_):
        r = snr(frame, (x_, y_), fwhm, full_output=True, exclude_negative_lobes=exclude_negative_lobes)
        return r[-1], r[2]
    if fmerit == "px":
        return one(y, x)
    yy, xx = disk_pixels(y, x, fwhm / 2.0)
    res = [one(y_, x_) for y_, x_ in zip(yy, xx)]
    snrs = np.array([r[0] for r in res])
    fluxes = np.array([r[1] for r in res])
    if fmerit == "max":
        return np.max(snrs), fluxes[int(np.argmax(snrs))]
    return np.mean(snrs), np.mean(fluxes)                   # 'mean'


def pca_grid(cube, angle_list, fwhm=None, range_pcs=None, source_xy=None, cube_ref=None, mode="fullfr",
             annulus_width=20, svd_mode="lapack", scaling=None, mask_center_px=None, fmerit="mean", collapse="median",
             ifs_collapse_range="all", verbose=True, full_output=False, debug=False, plot=True, save_plot=None,
             start_time=None, scale_list=None, initial_4dshape=None, weights=None, exclude_negative_lobes=False,
             **rot_options):
    """Grid of residual PCA frames for a range of numbers of PCs (reference utils_pca.py:25-428).

    Returns ``cubeout`` (``cubeout, pclist`` with ``full_output``); with ``source_xy``:
    ``(cubeout, finalfr, df, opt_npc)`` -- the frame of the best S/N, the pandas table PCs / S/Ns / fluxes and the
    optimal number of PCs.  ``plot`` / ``save_plot`` are accepted and ignored (no plotting on the accelerated path)."""
    torch = B.require_gpu()
    msdi = scale_list is not None and initial_4dshape is not None
    if msdi and mode != "fullfr":
        raise NotImplementedError("pca_grid on rescaled ADI+mSDI cubes: mode='fullfr' only")
    if not (isinstance(cube, np.ndarray) or B.is_device_tensor(cube)) or cube.ndim != 3:
        raise TypeError("Input cube is not a 3d array")
    dev_in = B.is_device_tensor(cube)
    out_dtype = None if dev_in else (cube.dtype if cube.dtype.kind == "f" else np.float64)
    n, ysz, xsz = cube.shape
    if source_xy is not None:
        if fwhm is None:
            raise ValueError("if source_xy is provided, so should fwhm")
        x, y = source_xy
    else:
        x = y = None
    if isinstance(range_pcs, list):
        pclist = list(range_pcs)
        pcmax = max(pclist)
    else:
        if range_pcs is None:
            pcmin, pcmax, step = 1, n - 1, 1
        elif len(range_pcs) == 2:
            pcmin, pcmax = range_pcs
            pcmax = min(pcmax, n)
            step = 1
        elif len(range_pcs) == 3:
            pcmin, pcmax, step = range_pcs
            pcmax = min(pcmax, n)
        else:
            raise TypeError("`range_pcs` must be None or a tuple, corresponding to (PC_INI, PC_MAX) or "
                            "(PC_INI, PC_MAX, STEP)")
        pclist = list(range(pcmin, pcmax + 1, step))
    if fmerit not in ("px", "max", "mean"):
        raise ValueError("Invalid value for fmerit: {}.".format(fmerit))
    collapse = _s(collapse)
    if collapse not in B.COLLAPSE_MODES or collapse == "stim":
        raise TypeError("mode not recognized")
    scaling = _s(scaling)
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=np.float64))
    if msdi:
        # `cube` is the frame-major stack of rescaled channels of a (z, n_adi, y, x) cube: every residual cube is de-scaled
        # and collapsed per multispectral frame (with `collapse`, as the reference does: utils_pca.py:201-220) first
        from ..preproc.rescaling import channel_operators, zoom_frames
        z4, n_adi, y4, x4 = initial_4dshape
        i0, i1 = (0, z4) if ifs_collapse_range == "all" else ifs_collapse_range
        zc = i1 - i0
        Einv = channel_operators(ysz, np.asarray(scale_list, dtype=np.float64)[i0:i1], inverse=True,
                                 out_size=max(y4, x4))
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    cube_t = B.to_device_f32(cube)
    ref_t = B.to_device_f32(cube_ref) if cube_ref is not None else None
    P = ysz * xsz
    pix_t = None
    if mode == "fullfr":
        M = _prep_fullfr(cube_t, scaling, mask_center_px)
        ref_lib = M if ref_t is None else _prep_fullfr(ref_t, scaling, mask_center_px)
    elif mode == "annular":
        if source_xy is None:
            raise TypeError("mode='annular' needs `source_xy` (the annulus passes through it)")
        y_cent, x_cent = frame_center(cube_t[0])
        ann_radius = dist(y_cent, x_cent, y, x)
        inrad = int(ann_radius - annulus_width / 2.0)
        outrad = int(ann_radius + annulus_width / 2.0)
        pix_t = torch.from_numpy(_annulus_pixels((ysz, xsz), inrad, outrad)).to(cube_t.device)
        M = _gather_annulus(cube_t, pix_t, scaling)
        ref_lib = M if ref_t is None else _gather_annulus(ref_t, pix_t, scaling)
    else:
        raise RuntimeError("Wrong mode. Choose either fullfr or annular")
    if pcmax > min(ref_lib.shape):
        msg = "{} PCs cannot be obtained from a matrix with size [{},{}]."
        msg += " Increase the size of the patches or request less PCs"
        raise RuntimeError(msg.format(pcmax, ref_lib.shape[0], ref_lib.shape[1]))
    from .svd import _decompose
    _sig, _E, V = _decompose(ref_lib, int(pcmax), want_pcs=True, leading_only=True)      # V: (pcmax, npx)
    coeff = B.cross_gram(M, V).to(torch.float32)                                        # M V^T: (n, pcmax)
    ctx = B.get_context(cube_t.device.index)
    npx = M.shape[1]
    with B.rotation_mode(rot_options.get("imlib", "vip-fft"), rot_options.get("interpolation", "lanczos4"),
                         rot_options.get("border_mode", "constant"), rot_options.get("mask_val")):
        frames = []
        for pc in pclist:
            C = coeff[:, :pc].contiguous()
            R = B.empty((n, npx), device=cube_t.device.index)
            ctx.call("vipmi_subtract_gemm_f32", B.ptr(M), B.ptr(C), B.ptr(V), n, int(pc), npx, B.ptr(R), None)
            if msdi:
                sel = R.reshape(n_adi, z4, ysz, xsz)[:, i0:i1].reshape(n_adi * zc, ysz, xsz).contiguous()
                desc = zoom_frames(sel, Einv, np.tile(np.arange(zc), n_adi))
                res_cube = B.collapse_batched(desc.reshape(n_adi, zc, desc.shape[1], desc.shape[2]), collapse)
            elif pix_t is None:
                res_cube = R.reshape(n, ysz, xsz)
            else:
                res_cube = torch.zeros((n, ysz, xsz), dtype=torch.float32, device=cube_t.device)
                ctx.call("vipmi_scatter_f32", B.ptr(R), n, P, B.ptr(pix_t), npx, B.ptr(res_cube))
            der = B.derotate(res_cube, angle_list, mask_nan=mv_nan, mask_zero=not mv_nan)
            frames.append(B.collapse(der, collapse, w=weights))
        cubeout = torch.stack(frames)

    def host(t):
        return t if dev_in else B.to_host(t, out_dtype)

    if x is not None and y is not None and fwhm is not None:
        frames_h = cubeout.cpu().numpy().astype(np.float64)
        snrlist, fluxlist = [], []
        for fr in frames_h:
            snr_value, flux = _get_snr(fr, y, x, fwhm, fmerit, exclude_negative_lobes)
            if np.isnan(snr_value):
                snr_value = 0
            snrlist.append(snr_value)
            fluxlist.append(flux)
        argmax = int(np.argmax(snrlist))
        opt_npc = pclist[argmax]
        from pandas import DataFrame
        df = DataFrame({"PCs": pclist, "S/Ns": snrlist, "fluxes": fluxlist})
        if debug:
            print(df, "\n")
        if verbose:
            print("Number of steps", len(pclist))
            print("Optimal number of PCs = {}, for S/N={:.3f}".format(opt_npc, snrlist[argmax]))
        finalfr = cubeout[argmax]
        frame_report(frames_h[argmax], fwhm, (x, y), verbose=verbose)
        return host(cubeout), host(finalfr), df, opt_npc
    if verbose:
        print("Computed residual frames for PCs interval: {}".format(range_pcs))
        print("Number of steps", len(pclist))
    if full_output:
        return host(cubeout), pclist
    return host(cubeout)


def pca_annulus(cube, angs, ncomp, annulus_width, r_guess, cube_ref=None, svd_mode="lapack", scaling=None,
                collapse="median", weights=None, collapse_ifs="mean", **rot_options):
    """PCA-ADI / PCA-RDI on one annulus of width ``annulus_width`` at radius ``r_guess`` (reference utils_pca.py:617-755):
    final frame (``collapse`` not None) or the cube of residuals (derotated if ``angs`` is not None), non-zero only on
    the annulus.  4-D cubes: one annulus PCA per channel, then ``collapse_ifs``."""
    torch = B.require_gpu()
    if not (isinstance(cube, np.ndarray) or B.is_device_tensor(cube)) or cube.ndim not in (3, 4):
        raise TypeError("`cube` must be a 3 or 4d array")
    dev_in = B.is_device_tensor(cube)
    out_dtype = None if dev_in else (cube.dtype if cube.dtype.kind == "f" else np.float64)
    scaling = _s(scaling)
    collapse = _s(collapse)
    if collapse is not None and (collapse not in B.COLLAPSE_MODES or collapse == "stim"):
        raise TypeError("mode not recognized")
    mask_val = rot_options.get("mask_val", np.nan)
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    angles = None if angs is None else check_pa_vector(np.asarray(angs, dtype=np.float64))

    def host(t, dt=None):
        return t if dev_in else B.to_host(t, dt or out_dtype)

    def one(cube_t, ref_t, k):
        n, ysz, xsz = cube_t.shape
        inrad = int(r_guess - annulus_width / 2.0)
        outrad = int(r_guess + annulus_width / 2.0)
        pix_t = torch.from_numpy(_annulus_pixels((ysz, xsz), inrad, outrad)).to(cube_t.device)
        data = _gather_annulus(cube_t, pix_t, scaling)
        data_svd = data if ref_t is None else _gather_annulus(ref_t, pix_t, scaling)
        k = int(k)
        if k > min(data_svd.shape):
            msg = "{} PCs cannot be obtained from a matrix with size [{},{}]."
            msg += " Increase the size of the patches or request less PCs"
            raise RuntimeError(msg.format(k, data_svd.shape[0], data_svd.shape[1]))
        res, _, _, _ = B.pca_project(data, k, ref=None if ref_t is None else data_svd)
        cube_zeros = torch.zeros((n, ysz, xsz), dtype=torch.float32, device=cube_t.device)
        ctx = B.get_context(cube_t.device.index)
        ctx.call("vipmi_scatter_f32", B.ptr(res.contiguous()), n, ysz * xsz, B.ptr(pix_t), int(pix_t.numel()),
                 B.ptr(cube_zeros))
        with B.rotation_mode(rot_options.get("imlib", "vip-fft"), rot_options.get("interpolation", "lanczos4"),
                             rot_options.get("border_mode", "constant"), rot_options.get("mask_val")):
            out = cube_zeros if angles is None else B.derotate(cube_zeros, angles, mask_nan=mv_nan, mask_zero=not mv_nan)
        if collapse is not None:
            return B.collapse(out, collapse, w=weights)
        return out

    cube_t = B.to_device_f32(cube)
    if cube.ndim == 3:
        ref_t = B.to_device_f32(cube_ref) if cube_ref is not None else None
        return host(one(cube_t, ref_t, ncomp))
    nch = cube.shape[0]
    refs = None
    if cube_ref is not None:
        if isinstance(cube_ref, (list, tuple)):
            refs = [B.to_device_f32(r) for r in cube_ref]
        elif cube_ref.ndim == 3:
            refs = [B.to_device_f32(cube_ref)] * nch
        else:
            refs = [B.to_device_f32(cube_ref[ch]) for ch in range(nch)]
    if np.isscalar(ncomp):
        ncomp = [ncomp] * nch
    elif isinstance(ncomp, list) and len(ncomp) != nch:
        raise TypeError("If ncomp is a list, in the case of a 4d input cube without input scale_list, it should have the "
                        "same length as the first dimension of the cube.")
    if collapse is None:
        raise ValueError("mode not supported. Provide value for collapse")
    frames = []
    for ch in range(nch):
        ref_ch = None
        if refs is not None:
            if refs[ch].ndim != 3:
                raise TypeError("Ref cube has wrong format for 4d input cube")
            ref_ch = refs[ch]
        frames.append(one(cube_t[ch], ref_ch, ncomp[ch]))
    ifs_res = torch.stack(frames)
    return host(B.collapse(ifs_res, _s(collapse_ifs)), np.float64)
