"""STIM maps: drop-in for ``vip_hci.metrics.stim`` (reference metrics/stim.py:24-118).

``stim_map`` is one pass of the column-reduction kernel (mean / population standard deviation over the frames,
float64 accumulation) followed by the reference's ``get_circle`` mask; the inverse / normalised maps add the FFT
derotation.  numpy in -> numpy out, cuda tensor in -> cuda tensor out.
"""
import numpy as np

from .. import backend as B
from ..var.coords import frame_center
from ..var.shapes import mask_circle


def _circle_keep_u8(n_y, n_x, radius):
    """1 where ``get_circle`` zeroes the map: (y-cy)^2 + (x-cx)^2 >= radius^2 (var/shapes.py:389-398)."""
    cy, cx = frame_center(np.zeros((n_y, n_x)))
    yy, xx = np.ogrid[:n_y, :n_x]
    inside = (yy - cy) ** 2 + (xx - cx) ** 2 < radius ** 2
    return (~inside).astype(np.uint8)


def _stim_dev(cube_t):
    torch = B._torch()
    t, n, nx = cube_t.shape
    det = B.collapse(cube_t, "stim")
    outside = torch.from_numpy(_circle_keep_u8(n, nx, int(np.round(n / 2.0)))).to(cube_t.device)
    return B.apply_mask(det.reshape(1, -1), outside.reshape(-1), 0.0).reshape(n, nx)


def _wrap(x, dev_in, like):
    if dev_in:
        return x
    out = x.cpu().numpy()
    return out.astype(like.dtype, copy=False) if like.dtype.kind == "f" else out.astype(np.float64)


def stim_map(cube_der):
    """mu / sigma over the frames of a de-rotated residual cube, inside the inscribed circle."""
    if cube_der.ndim != 3:
        raise TypeError("Input array is not a cube or 3d array")
    dev_in = B.is_device_tensor(cube_der)
    return _wrap(_stim_dev(B.to_device_f32(cube_der)), dev_in, cube_der)


def inverse_stim_map(cube, angle_list, **rot_options):
    """STIM map of the cube de-rotated with the opposite angles."""
    dev_in = B.is_device_tensor(cube)
    t = B.to_device_f32(cube)
    with B.rotation_mode(rot_options.get("imlib", "vip-fft"), rot_options.get("interpolation", "lanczos4"),
                         rot_options.get("border_mode", "constant"), rot_options.get("mask_val")):
        mask_val = rot_options.get("mask_val", np.nan)
        mv_nan = isinstance(mask_val, float) and bool(np.isnan(mask_val))
        der = B.derotate(t, -np.asarray(angle_list, dtype=np.float64), mask_nan=mv_nan, mask_zero=not mv_nan)
    return _wrap(_stim_dev(der), dev_in, cube)


def normalized_stim_map(cube, angle_list, mask=None, **rot_options):
    """STIM map divided by the maximum of the inverse STIM map (optionally outside a central mask)."""
    dev_in = B.is_device_tensor(cube)
    t = B.to_device_f32(cube)
    inv_map = inverse_stim_map(t, angle_list, **rot_options)
    if mask is not None:
        if np.isscalar(mask):
            inv_map = mask_circle(inv_map, mask)
        else:
            # binary mask (ones where the maximum may be taken): zero the rest
            torch = B._torch()
            off = torch.from_numpy((np.asarray(mask) == 0).astype(np.uint8)).to(inv_map.device)
            inv_map = B.apply_mask(inv_map.reshape(1, -1), off.reshape(-1), 0.0).reshape(inv_map.shape)
    # np.nanmax over the map = NaN-aware 'max' collapse of the pixels seen as a one-pixel cube
    max_inv = float(B.collapse(inv_map.reshape(-1, 1, 1), "max").item())
    if max_inv <= 0:
        raise ValueError("The normalization value is found to be {}".format(max_inv))
    with B.rotation_mode(rot_options.get("imlib", "vip-fft"), rot_options.get("interpolation", "lanczos4"),
                         rot_options.get("border_mode", "constant"), rot_options.get("mask_val")):
        der = B.derotate(t, np.asarray(angle_list, dtype=np.float64))
    return _wrap(B.lincomb(_stim_dev(der), None, 1.0 / max_inv), dev_in, cube)
