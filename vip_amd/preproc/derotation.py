"""cube_derotate / frame_rotate (imlib='vip-fft') on the device + the ADI index helpers (host).

Reference: preproc/derotation.py:51-328 (frame_rotate), :331-399 (cube_derotate), :410-496
(_find_indices_adi), :499-504 (_compute_pa_thresh), :507-539 (_define_annuli).
"""
import os

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
    return isinstance(mask_val, float) and bool(np.isnan(mask_val))


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
    if (not dev_in and mv_nan is not None and isinstance(array, np.ndarray) and array.dtype in (np.float32, np.float64)
            and array.nbytes >= (64 << 20) and array.shape[0] >= 32 and os.environ.get("VIPMI_HOSTIN", "1") != "0" and not B.is_async()):
        return _derotate_numpy_pipelined(array, angle_list, mv_nan, method, mask_val)
    t = B.to_device_f32(array)
    if mv_nan is None:
        out = B.rotate_interp(t, angle_list, str(getattr(interpolation, "value", interpolation)), cxy=cxy,
                              border_mode=border_mode)
    else:
        out = B.derotate(t, angle_list, mask_nan=mv_nan, mask_zero=not mv_nan, method=method, mask_val=mask_val)
    if dev_in:
        return out
    return B.to_host(out, array.dtype if array.dtype.kind == "f" else np.float64)


def _derotate_numpy_pipelined(array, angle_list, mv_nan, method, mask_val):
    """cube_derotate of a numpy cube: upload, rotation and download of a 400 x 512 x 512 cube take 8 + 3.8 + 8 ms one after the
    other, but the rotation is per frame and the link is full duplex -- blocks of frames go up on one copy stream (uploader
    thread), are rotated on the calling thread's stream, and come down on a second copy stream (downloader thread) straight
    into the result array, all three at once.  Same kernels on the same frames: bit-identical to the one-shot path."""
    import queue
    import threading
    torch = B.require_gpu()
    n = array.shape[0]
    out = np.empty(array.shape, dtype=array.dtype)
    nblk = max(2, min(8, n // 16))
    bounds = [round(b * n / nblk) for b in range(nblk + 1)]
    dev = torch.cuda.current_device()
    cur = torch.cuda.current_stream()
    s_in, s_out = B.side_streams(2, dev)
    s_in.wait_stream(cur)
    q_in, q_out = queue.Queue(maxsize=2), queue.Queue()
    err = []

    def uploader():
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(s_in):
                for b in range(nblk):
                    t_ = torch.from_numpy(np.ascontiguousarray(array[bounds[b]:bounds[b + 1]])).to(torch.device("cuda", dev))
                    if t_.dtype != torch.float32:
                        t_ = t_.to(torch.float32)          # (on the device: numpy's astype of a float64 cube costs more than the copy)
                    q_in.put((t_, s_in.record_event()))
        except BaseException as e:
            q_in.put(e)

    def downloader():
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(s_out):
                while True:
                    item = q_out.get()
                    if item is None:
                        return
                    r_, ev_, b0, b1 = item
                    s_out.wait_event(ev_)
                    dst = torch.from_numpy(out[b0:b1])
                    dst.copy_(r_ if dst.dtype == r_.dtype else r_.to(dst.dtype))     # (blocking: pageable destination)
        except BaseException as e:
            err.append(e)

    tu = threading.Thread(target=uploader, name="vipmi-upload", daemon=True)
    td = threading.Thread(target=downloader, name="vipmi-download", daemon=True)
    tu.start()
    td.start()
    try:
        for b in range(nblk):
            item = q_in.get()
            if isinstance(item, BaseException):
                raise item
            t_, ev = item
            cur.wait_event(ev)
            t_.record_stream(cur)
            r_ = B.derotate(t_, angle_list[bounds[b]:bounds[b + 1]], mask_nan=mv_nan, mask_zero=not mv_nan, method=method, mask_val=mask_val)
            r_.record_stream(s_out)
            q_out.put((r_, cur.record_event(), bounds[b], bounds[b + 1]))
            del t_, r_, item
    finally:
        q_out.put(None)
        td.join()
        while tu.is_alive():                 # (drain after an early exit: the uploader may be blocked on a full queue)
            try:
                q_in.get(timeout=0.05)
            except queue.Empty:
                pass
        tu.join()
    if err:
        raise err[0]
    return out


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


def _exclusion_windows(pa, thr, memo=None):
    """For every frame j the half-open index window [lo_j, hi_j) that the rotation criterion removes from its PCA
    library (reference derotation.py:444-461, as a matrix statement): with D_ji = |PA_j - PA_i| in float64,
    lo_j = the first i < j with D_ji < thr (j when there is none) and hi_j = the first i >= j with D_ji > thr (n when
    there is none).  Returns (lo, hi) as int arrays."""
    pa = np.asarray(pa)
    n = pa.shape[0]
    col = np.arange(n)
    if memo is not None and "D" in memo:                  # (one |PA_j - PA_i| matrix per angle list, not per annulus)
        D, before = memo["D"], memo["before"]
    else:
        D = np.abs(pa[:, None] - pa[None, :])
        before = col[None, :] < col[:, None]
        if memo is not None:
            memo["D"], memo["before"] = D, before
    close = (D < thr) & before
    lo = np.where(close.any(axis=1), close.argmax(axis=1), col)
    apart = (D > thr) & ~before
    hi = np.where(apart.any(axis=1), apart.argmax(axis=1), n)
    return lo, hi


def _library_of(pa, j, lo, hi, limit):
    """Frames outside [lo, hi), cut down to the ``limit`` frames nearest in parallactic angle when there are more (the
    reference's argsort on |dPA| decides ties: derotation.py:488-495).  ``limit`` None: no cut (int32 as the reference)."""
    n = pa.shape[0]
    keep = np.concatenate([np.arange(lo), np.arange(hi, n)])
    if limit is not None and keep.shape[0] > limit:
        nearest = np.argsort(np.abs(pa[keep] - pa[j]))[:limit]
        return np.sort(keep[nearest])
    return keep.astype("int32")


def _find_indices_adi(angle_list, frame, thr, nframes=None, out_closest=False, truncate=False,
                      max_frames=200):
    """Indices of the frames kept in the PCA library of ``frame`` (bit-exact index contract with the reference's
    derotation.py:410-496): row ``frame`` of ``_exclusion_windows``.  ``out_closest``: the last rejected frame on either
    side; ``nframes``: only nframes // 2 frames on either side of the window."""
    pa = np.asarray(angle_list)
    n = pa.shape[0]
    gap = np.abs(pa - pa[frame])
    near = np.flatnonzero(gap[:frame] < thr)
    lo = int(near[0]) if near.size else frame
    far = np.flatnonzero(gap[frame:] > thr)
    hi = frame + int(far[0]) if far.size else n
    if out_closest:
        return lo, hi - 1
    if nframes is not None:
        side = nframes // 2
        return np.concatenate([np.arange(max(lo - side, 0), lo), np.arange(hi, min(hi + side, n))]).astype("int32")
    return _library_of(pa, frame, lo, hi, min(n - 1, max_frames) if truncate else None)


def _find_indices_adi_all(angle_list, thr, truncate=False, max_frames=200, memo=None):
    """``[_find_indices_adi(angle_list, j, thr, truncate=truncate, max_frames=max_frames) for j in range(n)]`` from one
    |PA_j - PA_i| matrix (what the annular path calls: one plan per annulus, not n scans).
    ``memo`` (a dict the caller keeps for ONE angle list): the libraries depend on the threshold only through the integer
    exclusion windows, and neighbouring annuli -- whose thresholds differ by less than an angle step -- have the same windows:
    the list is then the same OBJECT (8 annuli of the benchmark's C3 call: 2 distinct sets; VIP's default asize = 4 on a
    512-px frame: 64 annuli).  The per-frame selections are ~8 us of host time each: 27 ms per new angle list at C3 without
    the memo, more than the device work of the call for the default annulus width."""
    pa = np.asarray(angle_list)
    n = pa.shape[0]
    lo, hi = _exclusion_windows(pa, thr, memo)
    limit = min(n - 1, max_frames) if truncate else None
    key = None
    if memo is not None:
        key = (lo.tobytes(), hi.tobytes(), limit)
        hit = memo.get(key)
        if hit is not None:
            return hit
    libs = [_library_of(pa, j, int(lo[j]), int(hi[j]), limit) for j in range(n)]
    if memo is not None:
        memo[key] = libs
    return libs


def _find_indices_adi_nframes_all(angle_list, thr, nframes):
    """``[_find_indices_adi(angle_list, j, thr, nframes=nframes) for j in range(n)]``: nframes // 2 frames on either side of every
    frame's exclusion window (median_sub's annular mode, medsub.py:602-676), from one |PA_j - PA_i| matrix."""
    pa = np.asarray(angle_list)
    n = pa.shape[0]
    lo, hi = _exclusion_windows(pa, thr)
    side = nframes // 2
    return [np.concatenate([np.arange(max(int(lo[j]) - side, 0), int(lo[j])), np.arange(int(hi[j]), min(int(hi[j]) + side, n))]).astype("int32")
            for j in range(n)]


def _compute_pa_thresh(ann_center, fwhm, delta_rot=1):
    return np.rad2deg(2 * np.arctan(delta_rot * fwhm / (2 * ann_center)))


def _define_annuli(angle_list, ann, n_annuli, fwhm, radius_int, annulus_width, delta_rot, n_segments,
                   verbose, strict=False):
    """Geometry and rotation threshold of annulus ``ann`` (reference derotation.py:507-539): the annulus
    starts at radius_int + ann * width -- one pixel earlier for the last one --, its centre is half a width further out,
    the threshold is the angle under which delta_rot * fwhm is seen from the centre, capped at 90 % of half the
    parallactic range (unless ``strict``, which only warns).  Returns (pa_threshold, inner_radius, ann_center)."""
    last = ann == n_annuli - 1
    inner_radius = radius_int + ann * annulus_width - (1 if last else 0)
    ann_center = inner_radius + annulus_width / 2
    pa_threshold = _compute_pa_thresh(ann_center, fwhm, delta_rot)
    half_range = np.abs(np.amax(angle_list) - np.amin(angle_list)) / 2
    cap = half_range - half_range * 0.1
    if pa_threshold >= cap:
        if not strict:
            print("PA threshold %.2f exceeds 90%% of half the rotation range: using %.2f" % (pa_threshold, float(cap)))
            pa_threshold = float(cap)
        elif int(verbose) > 1:
            print("WARNING: PA threshold %.2f of annulus %d exceeds %.2f (kept: strict)" % (pa_threshold, ann, float(cap)))
    if int(verbose):
        head = "Ann %d" % (ann + 1)
        if pa_threshold > 0:
            head += "    PA thresh: %5.2f" % pa_threshold
        print("%s    Ann center: %3.0f    N segments: %s " % (head, ann_center, n_segments))
    return pa_threshold, inner_radius, ann_center
