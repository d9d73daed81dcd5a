"""Host mirror of the var/shapes.py pieces on the PCA path (reference lines cited per function).

Index sets (disk / annulus membership) are computed on the host in float64 with exactly the
reference's expressions so that membership is bit-exact; the heavy array work (masking, scaling)
runs on the device.
"""
import functools

import numpy as np

from .. import backend as B
from .coords import frame_center


def disk_mask(shape, radius, cy=None, cx=None):
    """Boolean mask of skimage.draw.disk((cy, cx), radius, shape=shape): pixels with
    ((r-cy)/R)^2 + ((c-cx)/R)^2 < 1 (var/shapes.py:88)."""
    if cy is None or cx is None:
        cy, cx = int(shape[0] // 2), int(shape[1] // 2)
    rr, cc = np.mgrid[:shape[0], :shape[1]]
    return ((rr - cy) / radius) ** 2 + ((cc - cx) / radius) ** 2 < 1


def center_mask_u8(shape, radius):
    """uint8 pixel mask (1 = masked) reproducing ``mask_circle(cube3d, radius)``: for 3d/4d input the
    reference indexes ``[:, ind[1], ind[0]]`` (var/shapes.py:101), i.e. the transposed disk."""
    cy, cx = int(shape[0] // 2), int(shape[1] // 2)
    return np.ascontiguousarray(disk_mask(shape, radius, cy, cx).T).astype(np.uint8)


def mask_circle(array, radius, fillwith=0, mode="in", cy=None, cx=None, output="masked_arr"):
    """var/shapes.py:38-113.  numpy in -> numpy out; cuda tensor in -> cuda tensor out."""
    if not isinstance(fillwith, (int, float)):
        raise ValueError("`fillwith` must be integer, float or np.nan")
    shape = (array.shape[-2], array.shape[-1])
    if radius == 0:
        mask = np.ones(shape, dtype=bool) if mode == "in" else np.zeros(shape, dtype=bool)
        return mask if output == "bool_mask" else mask[0, 0] * array
    if cy is None or cx is None:
        cy, cx = frame_center(array)
    dm = disk_mask(shape, radius, cy, cx)
    if output == "bool_mask":
        return ~dm
    if output != "masked_arr":
        raise ValueError("output not recognized")
    # 3d/4d: transposed index order, as the reference
    m = dm if array.ndim == 2 else dm.T
    if mode == "out":
        m = ~m
    elif mode != "in":
        raise ValueError("mode not recognized")
    dev_in = B.is_device_tensor(array)
    t = B.to_device_f32(array)
    mk = B.to_device_f32(m.astype(np.float32)).to(dtype=B._torch().uint8)
    out = B.apply_mask(t.reshape(-1, shape[0] * shape[1]), mk.reshape(-1), float(fillwith)).reshape(t.shape)
    if dev_in:
        return out
    return out.cpu().numpy().astype(array.dtype, copy=False)


@functools.lru_cache(maxsize=8)
def _polar_grids(ny, nx):
    """Radius and azimuth (mod 2 pi) of every pixel about frame_center: the same float64 expressions as
    var/shapes.py:548-552, cached per frame shape (an annular PCA asks for them once per annulus)."""
    cy, cx = frame_center(np.zeros((ny, nx)))
    yy, xx = np.mgrid[:ny, :nx]
    rad = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
    phirot = np.arctan2(yy - cy, xx - cx) % (2 * np.pi)
    rad.setflags(write=False)
    phirot.setflags(write=False)
    return rad, phirot


def get_annulus_segments(data, inner_radius, width, nsegm=1, theta_init=0, optim_scale_fact=1,
                         mode="ind", out=False):
    """var/shapes.py:474-581 (host; index membership is bit-exact)."""
    if isinstance(data, tuple):
        array = np.zeros(data)
    else:
        array = np.asarray(data)
    if array.ndim != 2:
        raise TypeError("`data` must be a 2d array or a shape tuple")
    if not isinstance(nsegm, int):
        raise TypeError("`nsegm` must be an integer")
    azimuth_coverage = np.deg2rad(int(np.ceil(360 / nsegm)))
    twopi = 2 * np.pi
    rad, phirot = _polar_grids(array.shape[0], array.shape[1])
    outer_radius = inner_radius + (width * optim_scale_fact)
    if mode == "ind" and not out:
        # index sets only (what the annular plans ask for, once per annulus): the same float64 comparisons on the annulus' bounding
        # box instead of the whole frame, and no azimuth tests when one segment covers the full turn -- phirot lies in [0, 2 pi) and
        # deg2rad(360) is 2 pi exactly, so the reference's `(phirot >= 0) & (phirot < 2 pi)` holds everywhere
        cy, cx = frame_center(array)
        reach = int(np.ceil(outer_radius)) + 1
        y0, y1 = max(int(cy) - reach, 0), min(int(cy) + reach + 1, array.shape[0])
        x0, x1 = max(int(cx) - reach, 0), min(int(cx) + reach + 1, array.shape[1])
        rad_b, phi_b = rad[y0:y1, x0:x1], phirot[y0:y1, x0:x1]
        ring_b = (rad_b >= inner_radius) & (rad_b < outer_radius)
        res = []
        for i in range(nsegm):
            phi_start = np.deg2rad(theta_init) + (i * azimuth_coverage)
            phi_end = phi_start + azimuth_coverage
            if nsegm == 1 and phi_start == 0 and phi_end == twopi:
                m_b = ring_b
            elif phi_start < twopi and phi_end > twopi:
                m_b = (ring_b & (phi_b >= phi_start) & (phi_b <= twopi) | ring_b & (phi_b >= 0) & (phi_b < phi_end - twopi))
            elif phi_start >= twopi and phi_end > twopi:
                m_b = ring_b & (phi_b >= phi_start - twopi) & (phi_b < phi_end - twopi)
            else:
                m_b = ring_b & (phi_b >= phi_start) & (phi_b < phi_end)
            yy, xx = np.where(m_b)
            res.append((yy + y0, xx + x0))
        return res
    ring = (rad >= inner_radius) & (rad < outer_radius)
    masks = []
    for i in range(nsegm):
        phi_start = np.deg2rad(theta_init) + (i * azimuth_coverage)
        phi_end = phi_start + azimuth_coverage
        if phi_start < twopi and phi_end > twopi:
            masks.append(ring & (phirot >= phi_start) & (phirot <= twopi) |
                         ring & (phirot >= 0) & (phirot < phi_end - twopi))
        elif phi_start >= twopi and phi_end > twopi:
            masks.append(ring & (phirot >= phi_start - twopi) & (phirot < phi_end - twopi))
        else:
            masks.append(ring & (phirot >= phi_start) & (phirot < phi_end))
    if out:
        masks = list(~np.array(masks))
    if mode == "ind":
        return [np.where(mask) for mask in masks]
    elif mode == "val":
        return [array[mask] for mask in masks]
    elif mode == "mask":
        return [array * mask for mask in masks]
    raise ValueError("mode '{}' unknown!".format(mode))


def matrix_scaling(matrix, scaling):
    """var/shapes.py:740-781 (sklearn.preprocessing.scale semantics) on the device."""
    if scaling is None:
        return matrix
    if scaling not in ("temp-mean", "spat-mean", "temp-standard", "spat-standard"):
        raise ValueError("Scaling mode not recognized")
    dev_in = B.is_device_tensor(matrix)
    t = B.to_device_f32(matrix)
    out = B.scale(t, str(getattr(scaling, "value", scaling)))
    if dev_in:
        return out
    return out.cpu().numpy().astype(matrix.dtype if matrix.dtype.kind == "f" else np.float64, copy=False)


def prepare_matrix(array, scaling=None, mask_center_px=None, mode="fullfr", inner_radius=None,
                   outer_radius=None, discard_mask_pix=False, verbose=True):
    """var/shapes.py:784-873."""
    if mode == "annular":
        if inner_radius is None or outer_radius is None:
            raise ValueError("`inner_radius` and `outer_radius` must be defined in annular mode")
        fr_size = array.shape[1]
        annulus_width = int(np.round(outer_radius - inner_radius))
        ind = get_annulus_segments((fr_size, fr_size), inner_radius, annulus_width, nsegm=1)[0]
        yy, xx = ind
        matrix = matrix_scaling(array[:, yy, xx], scaling)
        if verbose:
            print("Done vectorizing the cube annulus. Matrix shape: ({}, {})".format(*matrix.shape))
        return matrix, ind
    elif mode == "fullfr":
        if discard_mask_pix:
            raise NotImplementedError("discard_mask_pix (left_eigv mode) is outside the accelerated path")
        if mask_center_px:
            array = mask_circle(array, mask_center_px)
        matrix = matrix_scaling(array.reshape(array.shape[0], -1), scaling)
        if verbose:
            print("Done vectorizing the frames. Matrix shape: ({}, {})".format(*matrix.shape))
        return matrix
    raise ValueError("mode not recognized")


def reshape_matrix(array, y, x):
    """var/shapes.py:876-910."""
    return array.reshape(array.shape[0], y, x)
