"""FFT rescaling of spectral channels (ADI+mSDI): drop-in for ``cube_rescaling_wavelengths`` / ``frame_rescaling`` /
``scale_fft`` with ``imlib='vip-fft'`` (reference preproc/rescaling.py:324-475, 506-682, 1114-1217).

The reference zooms every channel with two 2-D FFTs of awkward sizes (zero-pad to N + 2 kd, fft2, crop or pad the
spectrum to N + 2 kf, ifft2, real part, crop or pad back).  That chain is a separable LINEAR map

    Y = Re(E X E^T),    E = (select rows) . IDFT_(N+2kf) . (spectrum window) . DFT_(N+2kd) . (place at kd)

so here one small complex matrix E per scale factor is built on the host in float64 (closed form below, the reflect
padding of ``cube_rescaling_wavelengths`` and the final crops folded in) and the device applies it to all frames of
all channels with four real matrix-core products per frame (``vipmi_zoom_frames_f32``).  The reference evaluates
its forward FFT in float32 (numpy >= 2 keeps the input precision), so agreement is at the 1e-6 level, not 1e-15.
"""
import functools

import numpy as np

from .. import backend as B
from ..var.coords import frame_center


def _kd_kf(dim, scale):
    """The (kd, kf) pair scale_fft picks: N' = N + 2 kd, N'' = N + 2 kf with N''/N' closest to ``scale``
    (rescaling.py:1143-1161)."""
    kd_array = np.arange(dim / 2 + 1, dtype=int)
    yy = dim / 2 * (scale - 1) + kd_array.astype(float) * scale
    kf_array = np.round(yy).astype(int)
    imin = np.nanargmin(np.abs(yy - kf_array))
    return int(kd_array[imin]), int(kf_array[imin])


@functools.lru_cache(maxsize=512)
def _zoom_operator_even(dim, scale):
    """Complex E (dim x dim) with ``scale_fft(X, scale, ori_dim=True) == Re(E X E^T)`` for an even ``dim``.

    Per axis: y[m] = (1/N'') sum_{k=-K/2}^{K/2-1} e^{2 pi i k m / N''} sum_j x[j] e^{-2 pi i k (j + kd) / N'}, K = min(N', N'')
    (the frequencies that survive the crop / zero padding of the shifted spectrum).  The sum over k is a geometric
    series: with theta = m/N'' - (j + kd)/N' it equals e^{-i pi theta} sin(pi K theta) / sin(pi theta)."""
    if scale == 1:
        return np.eye(dim, dtype=complex)
    kd, kf = _kd_kf(dim, scale)
    dim_p, dim_pp = dim + 2 * kd, dim + 2 * kf
    K = min(dim_p, dim_pp)
    theta = np.arange(dim_pp)[:, None] / dim_pp - (np.arange(dim)[None, :] + kd) / dim_p
    den = np.sin(np.pi * theta)
    small = np.abs(den) < 1e-12
    ratio = np.where(small, float(K), np.sin(np.pi * K * theta) / np.where(small, 1.0, den))
    full = np.exp(-1j * np.pi * theta) * ratio / dim_pp            # (dim_pp, dim)
    E = np.zeros((dim, dim), dtype=complex)
    if dim_pp > dim:
        E[:] = full[kf:kf + dim]
    else:
        E[-kf:-kf + dim_pp] = full
    E.setflags(write=False)
    return E


def zoom_operator(dim, scale):
    """Complex E with ``frame_rescaling(X, scale=scale, imlib='vip-fft') == Re(E X E^T)`` for a dim x dim frame; odd
    frames are embedded at [1:, 1:] of an even one (rescaling.py:644-672)."""
    if scale is None:
        scale = 1.0
    if dim % 2:
        return _zoom_operator_even(dim + 1, float(scale))[1:, 1:]
    return _zoom_operator_even(dim, float(scale))


def _fold_reflect(A, size, pad):
    """A @ R for the reflect-padding matrix R ((size + 2 pad) x size, R X R^T == np.pad(X, pad, 'reflect')): the
    padded columns of A are folded back onto the columns they mirror."""
    if pad == 0:
        return np.array(A)
    E = np.array(A[:, pad:pad + size])
    E[:, 1:pad + 1] += A[:, pad - 1::-1][:, :pad]                       # padded column i (< pad) mirrors column pad - i
    E[:, size - 1 - pad:size - 1] += A[:, :pad + size - 1:-1][:, :pad]  # column pad+size+t mirrors column size-2-t
    return E


def padded_size(size, scal_list):
    """Frame size after the reflect padding of cube_rescaling_wavelengths (rescaling.py:431-440)."""
    max_sc = float(np.amax(scal_list))
    if max_sc > 1:
        new = int(np.ceil(max_sc * size))
        if (new - size) % 2 != 0:
            new += 1
        return new
    return size


def _square_crop(size_in, size, cy):
    """First row of ``get_square(frame, size, cy, cx)`` and the (parity-adjusted) size (var/shapes.py:302-338)."""
    if size_in % 2 == 0:
        if size % 2 != 0:
            size += 1
    elif size % 2 == 0:
        size += 1
    wing = (size - 1) / 2
    return int(cy - wing), size


def channel_operators(size, scal_list, inverse=False, out_size=None, crop_to=None):
    return _channel_operators(int(size), tuple(float(v) for v in np.asarray(scal_list, dtype=float)), bool(inverse),
                              None if out_size is None else int(out_size), None if crop_to is None else int(crop_to))


@functools.lru_cache(maxsize=16)
def _channel_operators(size, scal_list, inverse, out_size, crop_to):
    """Per-channel complex operators of ``cube_rescaling_wavelengths``.

    forward (``inverse=False``): frames of ``size`` -> reflect pad to ``padded_size`` -> zoom by s_c [-> centre crop to
    ``crop_to`` as ``cube_crop_frames`` does for ``crop_ifs``].
    inverse: frames of ``size`` -> zoom by 1/s_c -> ``get_square`` crop to ``out_size`` if smaller.
    Returns E of shape (nchan, dout, size)."""
    scal_list = np.asarray(scal_list, dtype=float)
    ops = []
    if not inverse:
        big = padded_size(size, scal_list)
        y0, dout = 0, big
        if crop_to is not None and crop_to != big:
            cy, _ = frame_center(np.zeros((big, big)))
            y0, dout = _square_crop(big, int(crop_to), cy)
        for s in scal_list:
            E = _fold_reflect(zoom_operator(big, s), size, (big - size) // 2)
            ops.append(E[y0:y0 + dout])
    else:
        y0, dout = 0, size
        if out_size is not None and size > out_size and float(np.amax(scal_list)) > 1:
            cy, _ = frame_center(np.zeros((size, size)))
            y0, dout = _square_crop(size, int(out_size), cy)
        for s in scal_list:
            ops.append(zoom_operator(size, 1.0 / s)[y0:y0 + dout])
    return np.stack(ops)


def _upload_ops(E, device):
    """(Er, Ei) float32 cuda tensors [nchan][dout][ldk], ldk = din rounded up to 4, zero padded."""
    torch = B._torch()
    nch, dout, din = E.shape
    ldk = (din + 3) // 4 * 4
    er = np.zeros((nch, dout, ldk), dtype=np.float32)
    ei = np.zeros((nch, dout, ldk), dtype=np.float32)
    er[:, :, :din] = E.real
    ei[:, :, :din] = E.imag
    return torch.from_numpy(er).to(device), torch.from_numpy(ei).to(device), ldk


def zoom_frames(X, E, chan):
    """X: (nb, din, din) float32 cuda tensor; E: (nchan, dout, din) complex host array; chan: (nb,) channel of every
    frame.  Returns (nb, dout, dout) float32 cuda tensor = Re(E_c X E_c^T)."""
    torch = B._torch()
    nb, din, _ = X.shape
    nch, dout, din2 = E.shape
    if din2 != din:
        raise ValueError("operator / frame size mismatch")
    er, ei, ldk = _upload_ops(E, X.device)
    chan_t = torch.from_numpy(np.ascontiguousarray(chan, dtype=np.int32)).to(X.device)
    work = B.empty((2 * nb * dout * ldk,), device=X.device.index)
    out = B.empty((nb, dout, dout), device=X.device.index)
    ctx = B.get_context(X.device.index)
    X = X.contiguous()
    ctx.call("vipmi_zoom_frames_f32", B.ptr(X), nb, din, B.ptr(er), B.ptr(ei), B.ptr(chan_t), dout, ldk,
             B.ptr(work), B.ptr(out))
    return out


def cube_rescaling_wavelengths(cube, scal_list, full_output=True, inverse=False, y_in=None, x_in=None,
                               imlib="vip-fft", interpolation="lanczos4", collapse="median", pad_mode="reflect",
                               nproc=1):
    """Rescale the channels of a (n_channels, y, x) cube by ``scal_list`` about the frame centre (``inverse``: by
    1/scal_list, then crop to (y_in, x_in)).  Same returns as the reference: ``frame`` or
    ``(cube, frame, y, x, cy, cx)``."""
    if str(getattr(imlib, "value", imlib)) != "vip-fft":
        raise NotImplementedError("vip_amd implements imlib='vip-fft' only")
    if pad_mode != "reflect":
        raise NotImplementedError("only pad_mode='reflect' is accelerated")
    if cube.ndim != 3:
        raise TypeError("Input array is not a cube or 3d array")
    n, y, x = cube.shape
    if y != x:
        raise ValueError("FFT scaling only supports square input arrays")
    scal_list = np.asarray(scal_list, dtype=float)
    dev_in = B.is_device_tensor(cube)
    t = B.to_device_f32(cube)
    if inverse:
        if float(np.amax(scal_list)) > 1 and (y_in is None or x_in is None):
            raise ValueError("Provide y_in and x_in when inverse=True")
        siz = max(y_in, x_in) if y_in is not None else None
        E = channel_operators(y, scal_list, inverse=True, out_size=siz)
        big_y = y
        cy, cx = frame_center(np.zeros((y, x)))
    else:
        E = channel_operators(y, scal_list)
        big_y = E.shape[1]
        cy, cx = frame_center(np.zeros((big_y, big_y)))
    out = zoom_frames(t, E, np.arange(n))
    frame = B.collapse(out, str(getattr(collapse, "value", collapse)))

    def host(v):
        return v if dev_in else v.cpu().numpy().astype(np.float64)

    if full_output:
        return host(out), host(frame), big_y, big_y, cy, cx
    return host(frame)
