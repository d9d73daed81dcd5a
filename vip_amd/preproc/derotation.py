"""cube_derotate / frame_rotate (imlib='vip-fft') on the device + the ADI index helpers (host).

Reference: preproc/derotation.py:51-328 (frame_rotate), :331-399 (cube_derotate), :410-496
(_find_indices_adi), :499-504 (_compute_pa_thresh), :507-539 (_define_annuli).
"""
import numpy as np

from .. import backend as B


def _check_rot_options(imlib, cxy, edge_blend, mask_val, shape, interpolation="lanczos4", border_mode="constant"):
    imlib, interpolation = B.check_imlib(imlib, interpolation)
    if edge_blend not in (None, ""):
        raise NotImplementedError("edge_blend is outside the accelerated path")
    if imlib == "opencv":
        B.check_border(border_mode)
        return None
    if cxy is not None:
        cx, cy = cxy
        if (cy, cx) != (shape[0] // 2, shape[1] // 2):
            raise ValueError("'vip-fft' imlib does not yet allow for custom center to be  provided ")
    mv_nan = isinstance(mask_val, float) and np.isnan(mask_val)
    if not mv_nan and mask_val != 0:
        raise NotImplementedError("mask_val must be np.nan or 0 on the device path")
    return mv_nan


def cube_derotate(array, angle_list, imlib="vip-fft", interpolation="lanczos4", cxy=None, nproc=1,
                  border_mode="constant", mask_val=np.nan, edge_blend=None, interp_zeros=False, ker=1,
                  method="auto"):
    """Rotate frame i by -angle_list[i] degrees: the reference's 3-shear FFT rotation (``imlib='vip-fft'``, the
    parity path) or OpenCV's interpolating warpAffine (``imlib='opencv'``, ``interpolation`` = 'nearneig' | 'bilinear'
    | 'bicubic' | 'lanczos4', ``border_mode`` = 'constant' | 'edge' | 'symmetric' | 'reflect' | 'wrap', optional centre ``cxy``; derotation.py:279-305).

    Output dtype follows the reference's ``nproc=1`` branch (same dtype as the input);
    ``nproc`` is accepted and ignored (all frames are rotated concurrently on the GPU).
    ``method``: 'auto' | 'fft' | 'direct' (device algorithm of 'vip-fft', identical results up to float32 rounding)."""
    if array.ndim != 3:
        raise TypeError("Input array is not a cube or 3d array.")
    mv_nan = _check_rot_options(imlib, cxy, edge_blend, mask_val, array.shape[1:], interpolation, border_mode)
    angle_list = np.asarray(angle_list, dtype=np.float64)
    if angle_list.shape[0] != array.shape[0]:
        raise ValueError("`angle_list` must have one angle per frame")
    dev_in = B.is_device_tensor(array)
    t = B.to_device_f32(array)
    if mv_nan is None:
        out = B.rotate_interp(t, angle_list, str(getattr(interpolation, "value", interpolation)), cxy=cxy,
                              border_mode=border_mode)
    else:
        out = B.derotate(t, angle_list, mask_nan=mv_nan, mask_zero=not mv_nan, method=method)
    if dev_in:
        return out
    return out.cpu().numpy().astype(array.dtype if array.dtype.kind == "f" else np.float64, copy=False)


def frame_rotate(array, angle, imlib="vip-fft", interpolation="lanczos4", cxy=None,
                 border_mode="constant", mask_val=np.nan, edge_blend=None, interp_zeros=False, ker=1,
                 method="auto"):
    """Rotate one frame by ``angle`` degrees (the reference returns float64 for 'vip-fft' and cv2's float32 for
    'opencv'; so do we for numpy input)."""
    if array.ndim != 2:
        raise TypeError("Input array is not a frame or 2d array")
    dev_in = B.is_device_tensor(array)
    res = cube_derotate(array[None], np.array([-float(angle)]), imlib=imlib, interpolation=interpolation, cxy=cxy,
                        border_mode=border_mode, mask_val=mask_val, edge_blend=edge_blend, method=method)[0]
    if dev_in:
        return res
    return res.astype(np.float32 if str(getattr(imlib, "value", imlib)) == "opencv" else np.float64)


def _find_indices_adi(angle_list, frame, thr, nframes=None, out_closest=False, truncate=False,
                      max_frames=200):
    """Indices of the frames kept in the PCA library of ``frame`` (bit-exact index contract)."""
    n = angle_list.shape[0]
    index_prev = 0
    index_foll = frame
    for i in range(0, frame):
        if np.abs(angle_list[frame] - angle_list[i]) < thr:
            index_prev = i
            break
        else:
            index_prev += 1
    for k in range(frame, n):
        if np.abs(angle_list[k] - angle_list[frame]) > thr:
            index_foll = k
            break
        else:
            index_foll += 1
    if out_closest:
        return index_prev, index_foll - 1
    if nframes is not None:
        window = nframes // 2
        ind1 = max(index_prev - window, 0)
        ind4 = min(index_foll + window, n)
        return np.array(list(range(ind1, index_prev)) + list(range(index_foll, ind4)), dtype="int32")
    half1 = range(0, index_prev)
    half2 = range(index_foll, n)
    indices = np.array(list(half1) + list(half2), dtype="int32")
    if truncate:
        thr = min(n - 1, max_frames)
        all_indices = np.array(list(half1) + list(half2))
        if len(all_indices) > thr:
            dPA = np.abs(angle_list[all_indices] - angle_list[frame])
            indices = np.sort(all_indices[np.argsort(dPA)][:thr])
    return indices


def _find_indices_adi_all(angle_list, thr, truncate=False, max_frames=200):
    """``[_find_indices_adi(angle_list, j, thr, truncate=truncate, max_frames=max_frames) for j in range(n)]``
    with the two scans done for all frames at once on the |PA_j - PA_i| matrix (same float64 comparisons); the
    truncation keeps the reference's per-frame ``np.argsort`` call so that ties break identically."""
    a = np.asarray(angle_list)
    n = a.shape[0]
    D = np.abs(a[:, None] - a[None, :])
    idx = np.arange(n)
    below = (D < thr) & (idx[None, :] < idx[:, None])            # candidates i < j of the first scan
    prev = np.where(below.any(axis=1), below.argmax(axis=1), idx)
    above = (D > thr) & (idx[None, :] >= idx[:, None])           # candidates k >= j of the second scan
    foll = np.where(above.any(axis=1), above.argmax(axis=1), n)
    lim = min(n - 1, max_frames) if truncate else n
    out = []
    for j in range(n):
        all_indices = np.concatenate([idx[:prev[j]], idx[foll[j]:]])
        if truncate and all_indices.shape[0] > lim:
            dPA = np.abs(a[all_indices] - a[j])
            out.append(np.sort(all_indices[np.argsort(dPA)][:lim]))
        else:
            out.append(all_indices.astype("int32"))
    return out


def _compute_pa_thresh(ann_center, fwhm, delta_rot=1):
    return np.rad2deg(2 * np.arctan(delta_rot * fwhm / (2 * ann_center)))


def _define_annuli(angle_list, ann, n_annuli, fwhm, radius_int, annulus_width, delta_rot, n_segments,
                   verbose, strict=False):
    verbosity = int(verbose)
    if ann == n_annuli - 1:
        inner_radius = radius_int + (ann * annulus_width - 1)
    else:
        inner_radius = radius_int + ann * annulus_width
    ann_center = inner_radius + (annulus_width / 2)
    pa_threshold = _compute_pa_thresh(ann_center, fwhm, delta_rot)
    mid_range = np.abs(np.amax(angle_list) - np.amin(angle_list)) / 2
    if pa_threshold >= mid_range - mid_range * 0.1:
        new_pa_th = float(mid_range - mid_range * 0.1)
        if strict:
            if verbosity > 1:
                print("WARNING: PA threshold {:.2f} is too big, recommended  value for annulus {:.0f}: "
                      "{:.2f}".format(pa_threshold, ann, new_pa_th))
        else:
            print("PA threshold {:.2f} is likely too big, will be set to {:.2f}".format(pa_threshold, new_pa_th))
            pa_threshold = new_pa_th
    if verbosity:
        if pa_threshold > 0:
            print("Ann {}    PA thresh: {:5.2f}    Ann center: {:3.0f}    N segments: {} ".format(
                ann + 1, pa_threshold, ann_center, n_segments))
        else:
            print("Ann {}    Ann center: {:3.0f}    N segments: {} ".format(ann + 1, ann_center, n_segments))
    return pa_threshold, inner_radius, ann_center
