"""ADI+mSDI full-frame PCA (4-D cube + ``scale_list``): device versions of ``_adimsdi_doublepca`` /
``_adimsdi_doublepca_ifs`` (reference psfsub/pca_fullfr.py:1263-1549) and ``_adimsdi_singlepca`` (:1038-1216).

The spectral channels are rescaled with the matrix-core zoom of ``vip_amd.preproc.rescaling`` (all frames of all
channels in one call), the PCA stages reuse the full-frame kernels, the de-scaled channels are collapsed with the
collapse kernel.  Single-pass mode also takes a 4-D ``cube_ref`` and a tuple / list ``ncomp`` (grid of frames, with S/N
scoring at ``source_xy``).  Double-pass mode takes ``cube_ref`` (RSDI) and a rotation threshold at ``source_xy`` as well.  Not accelerated (NotImplementedError):
``batch``, ``mask_rdi``, ``smooth_first_pass``, ``imlib2`` other than 'vip-fft'.
"""
import numpy as np

from .. import backend as B
from ..preproc.parangles import check_pa_vector
from ..preproc.rescaling import channel_operators, zoom_frames
from ..var.shapes import center_mask_u8


def _s(x):
    return str(getattr(x, "value", x)) if x is not None else None


def _prep(M3, scaling, mask_center_px):
    """prepare_matrix(mode='fullfr') of a device cube (nf, y, x) -> (nf, P)."""
    nf, y, x = M3.shape
    m = M3.reshape(nf, -1)
    if mask_center_px:
        mask = B.to_device_f32(center_mask_u8((y, x), mask_center_px).astype(np.float32)).to(B._torch().uint8)
        m = B.apply_mask(m, mask.reshape(-1), 0.0)
    if scaling is not None:
        m = B.scale(m, scaling)
    return m


def _residuals(M3, ncomp, scaling, mask_center_px):
    """_project_subtract(cube, None, ncomp, scaling, mask_center_px, ...) -> residual cube (nf, y, x)."""
    nf, y, x = M3.shape
    M = _prep(M3, scaling, mask_center_px)
    if ncomp > min(M.shape):
        msg = "{} PCs cannot be obtained from a matrix with size [{},{}]."
        msg += " Increase the size of the patches or request less PCs"
        raise RuntimeError(msg.format(ncomp, M.shape[0], M.shape[1]))
    res = B.pca_project(M, int(ncomp))[0]
    return res.reshape(nf, y, x)


def _residuals_batched(M4, ncomp, scaling, mask_center_px):
    """``_residuals`` for a stack of equally shaped cubes, M4 = (batch, nf, y, x): the batch of small decompositions
    (one spectral PCA per multispectral frame in the first pass of ADI+mSDI, pca_fullfr.py:1482-1520) shares ONE Gram
    launch and ONE eigensolver launch (a workgroup per problem); only the two projection products stay per cube."""
    torch = B._torch()
    nb, nf, y, x = M4.shape
    P = y * x
    if ncomp > min(nf, P):
        msg = "{} PCs cannot be obtained from a matrix with size [{},{}]."
        msg += " Increase the size of the patches or request less PCs"
        raise RuntimeError(msg.format(ncomp, nf, P))
    if scaling is None and not mask_center_px:
        M = M4.reshape(nb, nf, P)
    else:
        M = torch.stack([_prep(M4[b], scaling, mask_center_px) for b in range(nb)])
    G = B.gram_batched(M)
    ev, ec = B.eigh_topk(G, int(ncomp))                                   # (nb, k), (nb, k, nf)
    keep = (ev > ev[:, :1] * 1e-12).to(torch.float32)
    E = (ec.to(torch.float32) * keep[:, :, None]).contiguous()            # rows = leading eigenvectors
    R = B.project_batched(M.contiguous(), E)                              # every frame's projection in two launches
    return R.reshape(nb, nf, y, x)


def _check(cube, angle_list, scale_list):
    z, n, y_in, x_in = cube.shape
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=np.float64))
    if not angle_list.shape[0] == n:
        raise ValueError("Angle list vector has wrong length. It must equal the number frames in the cube")
    if scale_list is None:
        raise ValueError("Scaling factors vector must be provided")
    scale_list = np.asarray(scale_list, dtype=np.float64)
    if scale_list.ndim > 1:
        raise ValueError("Scaling factors vector is not 1d")
    if not scale_list.shape[0] == z:
        raise ValueError("Scaling factors vector has wrong length")
    if y_in != x_in:
        raise ValueError("FFT scaling only supports square input arrays")
    return angle_list, scale_list


def _frame_major(cube4):
    """(z, n, y, x) -> (n*z, y, x) with index f*z + c, the order of the reference's ``big_cube``."""
    z, n, y, x = cube4.shape
    return cube4.permute(1, 0, 2, 3).reshape(n * z, y, x).contiguous()


def adimsdi_double(cube, angle_list, scale_list, ncomp, scaling, mask_center_px, collapse, collapse_ifs,
                   ifs_collapse_range, weights, mv_nan, verbose, cube_ref=None, ref_strategy="RSDI", source_xy=None,
                   delta_rot=None, fwhm=4, min_frames_pca=10, max_frames_pca=None, cube_sig=None):
    """Returns (res_cube_channels (n + nr, y, x), residuals_cube_channels_ (n, y, x), frame) as device tensors.
    ``cube_ref`` (z, nr, y, x): its multispectral frames go through the first (spectral) stage with the science frames
    and form the library of the second stage (pca_fullfr.py:1279-1283,1388-1400) -- or join every frame's library when a
    rotation threshold is applied at ``source_xy`` (:1403-1459).  ``cube_sig`` (n, y, x): estimate of the signal in the
    frames the SECOND stage works on, handed to its ``_project_subtract`` (:1395,1409,1447)."""
    torch = B._torch()
    z, n, y_in, x_in = cube.shape
    if not isinstance(ncomp, tuple):
        raise TypeError("`ncomp` must be a tuple when a double pass PCA is performed")
    ncomp_ifs, ncomp_adi = ncomp
    angle_list, scale_list = _check(cube, angle_list, scale_list)
    nr = 0
    if cube_ref is not None:
        if cube_ref.ndim != 4 or cube_ref.shape[0] != z or tuple(cube_ref.shape[2:]) != (y_in, x_in):
            raise TypeError("Ref cube has wrong format for 4d input cube")
        nr = int(cube_ref.shape[1])
        cube = torch.cat((cube, cube_ref), dim=1)
    n_sci, n = n, n + nr                         # every multispectral frame, science first (pca_fullfr.py:1279-1281)
    if type(scaling) is not tuple:
        scaling = (scaling, scaling)
    if verbose:
        print("{} spectral channels in IFS cube".format(z))
    if ncomp_ifs is not None and ncomp_ifs > z:
        ncomp_ifs = min(ncomp_ifs, z)
        print("Number of PCs too high (max PCs={}), using {} PCs instead".format(z, ncomp_ifs))
    i0, i1 = (0, z) if ifs_collapse_range == "all" else ifs_collapse_range
    zc = i1 - i0
    if ncomp_ifs is None:
        # first stage skipped: median of the channels of every multispectral frame (pca_fullfr.py:1479-1480)
        per = cube[i0:i1].permute(1, 0, 2, 3).contiguous()               # (n, zc, y, x)
        res_cube_channels = B.collapse_batched(per, "median")
    else:
        # 1. rescale every channel of every frame (reflect padded to the largest scale), frame-major order
        E = channel_operators(y_in, scale_list)
        big = E.shape[1]
        resc = zoom_frames(_frame_major(cube), E, np.tile(np.arange(z), n))          # (n*z, big, big)
        # 2. per multispectral frame: PCA over the z channels
        res_all = _residuals_batched(resc.reshape(n, z, big, big), int(ncomp_ifs), scaling[0], mask_center_px)
        if (i0, i1) != (0, z):
            res_all = res_all[:, i0:i1].contiguous()
        # 3. de-scale, crop back to the input size, collapse the channels
        Einv = channel_operators(big, scale_list[i0:i1], inverse=True, out_size=max(y_in, x_in))
        desc = zoom_frames(res_all.reshape(n * zc, big, big), Einv, np.tile(np.arange(zc), n))
        ys = desc.shape[1]
        desc = desc.reshape(n, zc, ys, ys)
        res_cube_channels = B.collapse_batched(desc, _s(collapse_ifs))
        if mask_center_px:
            mask = B.to_device_f32(center_mask_u8((ys, ys), mask_center_px).astype(np.float32)).to(torch.uint8)
            res_cube_channels = B.apply_mask(res_cube_channels.reshape(n, -1), mask.reshape(-1), 0.0).reshape(n, ys, ys)
    if ncomp_adi is None:
        if verbose:
            print("{} ADI frames".format(n_sci))
            print("De-rotating and combining frames (skipping PCA)")
        src = res_cube_channels[:n_sci]
    else:
        if ncomp_adi > n:
            ncomp_adi = n
            print("Number of PCs too high, using  maximum of {} PCs instead".format(n_sci))
        if verbose:
            print("{} ADI frames".format(n_sci))
            if nr:
                print("+ {} reference frames".format(nr))
            print("Second PCA stage exploiting rotational variability")
        sci, ref = res_cube_channels[:n_sci], (res_cube_channels[n_sci:] if nr else None)
        ys = res_cube_channels.shape[-1]
        if source_xy is not None:
            # rotation threshold at source_xy: per-frame libraries (+ every reference frame), pca_fullfr.py:1403-1459
            from .pca_fullfr import _pca_pa_rejection
            R, _M, _ln = _pca_pa_rejection(sci.contiguous(), angle_list, int(ncomp_adi), source_xy, delta_rot, fwhm,
                                           scaling[1], mask_center_px, min_frames_pca, max_frames_pca, False,
                                           cube_sig=cube_sig, cube_ref=None if ref is None else ref.contiguous())
            src = R.reshape(n_sci, ys, ys)
        elif nr and "A" in str(ref_strategy):
            # the reference projects all n + nr frames here and then de-rotates them with the n angles (:1388-1399,
            # :1462-1465): an IndexError in cube_derotate.  Said clearly instead.
            raise IndexError("ADI+mSDI double pass: ref_strategy %r with cube_ref needs source_xy (the reference de-rotates "
                             "%d residual frames with %d angles here)" % (ref_strategy, n, n_sci))
        elif nr:
            M = _prep(sci.contiguous(), scaling[1], mask_center_px)
            Mr = _prep(ref.contiguous(), scaling[1], mask_center_px)
            if ncomp_adi > min(Mr.shape):
                msg = "{} PCs cannot be obtained from a matrix with size [{},{}]."
                msg += " Increase the size of the patches or request less PCs"
                raise RuntimeError(msg.format(ncomp_adi, Mr.shape[0], Mr.shape[1]))
            if cube_sig is not None:                                # (PCs from the reference frames; M - S is projected, S added back)
                from .pca_fullfr import _project_subtract
                src = _project_subtract(sci.contiguous(), ref.contiguous(), int(ncomp_adi), scaling[1], mask_center_px, "lapack",
                                        False, False, cube_sig_t=cube_sig)
            else:
                src = B.pca_project(M, int(ncomp_adi), ref=Mr)[0].reshape(n_sci, ys, ys)
        elif cube_sig is not None:
            from .pca_fullfr import _project_subtract
            src = _project_subtract(res_cube_channels.contiguous(), None, int(ncomp_adi), scaling[1], mask_center_px, "lapack",
                                    False, False, cube_sig_t=cube_sig)
        else:
            src = _residuals(res_cube_channels, int(ncomp_adi), scaling[1], mask_center_px)
    der = B.derotate(src, angle_list, mask_nan=mv_nan, mask_zero=not mv_nan)
    frame = B.collapse(der, _s(collapse), w=weights)
    return res_cube_channels, der, frame


def adimsdi_single(cube, angle_list, scale_list, ncomp, scaling, mask_center_px, collapse, collapse_ifs,
                   ifs_collapse_range, crop_ifs, weights, mv_nan, verbose, cube_ref=None, grid_args=None):
    """int ``ncomp``: returns (cube_allfr_residuals (z*n, Y, X), cube_desc_residuals (zc, n, y, x), cube_adi_residuals
    (n, y, x), frame) as device tensors; ``cube_ref`` (4-D, same channels): rescaled like the cube, its frames are the
    PCA library (pca_fullfr.py:1099-1115).  Tuple / list ``ncomp``: the grid of final frames through
    ``utils_pca.pca_grid`` (one decomposition, every residual cube de-scaled, collapsed, derotated; pca_fullfr.py:1202-1236)
    -- returns what ``pca_grid`` returns; ``grid_args``: fwhm, source_xy, full_output, rot_options."""
    torch = B._torch()
    z, n, y_in, x_in = cube.shape
    angle_list, scale_list = _check(cube, angle_list, scale_list)
    if isinstance(ncomp, (float, np.floating)):
        raise NotImplementedError("float ncomp (CEVR) in single-pass ADI+mSDI is not accelerated")
    if not np.isscalar(ncomp) and not isinstance(ncomp, (tuple, list)):
        raise TypeError("`ncomp` must be an int, float, tuple or list for single-pass PCA")
    if verbose:
        print("Rescaling the spectral channels to align the speckles")
    E = channel_operators(y_in, scale_list, crop_to=y_in if crop_ifs else None)
    big_cube = zoom_frames(_frame_major(cube), E, np.tile(np.arange(z), n))           # (n*z, Y, Y)
    Y = big_cube.shape[1]
    big_ref = None
    if cube_ref is not None:
        if cube_ref.dim() != 4 or cube_ref.shape[0] != z or tuple(cube_ref.shape[2:]) != (y_in, x_in):
            raise TypeError("Ref cube has wrong format for 4d input cube")
        if verbose:
            print("Rescaling the spectral channels of the reference cube..")
        big_ref = zoom_frames(_frame_major(cube_ref), E, np.tile(np.arange(z), cube_ref.shape[1]))
    if verbose:
        print("{} total frames".format(n * z))
        print("Performing single-pass PCA")
    if not np.isscalar(ncomp):
        from .utils_pca import pca_grid
        ga = dict(grid_args or {})
        rot_options = ga.pop("rot_options", {})
        return pca_grid(big_cube, angle_list, ga.get("fwhm"), range_pcs=ncomp, source_xy=ga.get("source_xy"),
                        cube_ref=None, mode="fullfr", scaling=scaling, mask_center_px=mask_center_px, fmerit="mean",
                        collapse=collapse, ifs_collapse_range=ifs_collapse_range, verbose=verbose,
                        full_output=ga.get("full_output", False), scale_list=scale_list,
                        initial_4dshape=tuple(cube.shape), weights=weights, **rot_options)
    if big_ref is None:
        res_cube = _residuals(big_cube, int(ncomp), scaling, mask_center_px)
    else:
        M = _prep(big_cube, scaling, mask_center_px)
        R = _prep(big_ref, scaling, mask_center_px)
        if ncomp > min(R.shape):
            msg = "{} PCs cannot be obtained from a matrix with size [{},{}]."
            msg += " Increase the size of the patches or request less PCs"
            raise RuntimeError(msg.format(ncomp, R.shape[0], R.shape[1]))
        res_cube = B.pca_project(M, int(ncomp), ref=R)[0].reshape(n * z, Y, Y)
    i0, i1 = (0, z) if ifs_collapse_range == "all" else ifs_collapse_range
    zc = i1 - i0
    sel = res_cube.reshape(n, z, Y, Y)[:, i0:i1].reshape(n * zc, Y, Y).contiguous()
    Einv = channel_operators(Y, scale_list[i0:i1], inverse=True, out_size=max(y_in, x_in))
    desc = zoom_frames(sel, Einv, np.tile(np.arange(zc), n))
    ys = desc.shape[1]
    desc = desc.reshape(n, zc, ys, ys)
    resadi = B.collapse_batched(desc, _s(collapse_ifs))
    cube_desc = desc.permute(1, 0, 2, 3).contiguous()
    der = B.derotate(resadi, angle_list, mask_nan=mv_nan, mask_zero=not mv_nan)
    if mask_center_px:
        mask = B.to_device_f32(center_mask_u8((ys, ys), mask_center_px).astype(np.float32)).to(torch.uint8)
        der = B.apply_mask(der.reshape(n, -1), mask.reshape(-1), 0.0).reshape(n, ys, ys)
    frame = B.collapse(der, _s(collapse), w=weights)
    return res_cube, cube_desc, resadi, frame
