"""S/N of a test resolution element (reference metrics/snr_source.py:226-456,515-600), host side.

Only what the S/N-scored PCA grid needs (``psfsub/utils_pca.py:239-280``): ``snr``, ``indep_ap_centers``,
``frame_report`` for given positions.  The reference sums the apertures with photutils'
``aperture_photometry(method='exact')`` (photutils 2.3.0 in the reference's lock file; not in this image): every pixel
counts with the exact area of its unit square inside the circle.  ``aperture_sums_exact`` restates that geometry in
closed form; the numbers are small host work on ONE final frame per grid entry, so nothing here touches the GPU.
"""
import numpy as np

from ..var.coords import frame_center
from ..var.shapes import disk_mask


def _quadrant_area(x, y, r):
    """Area of {0 <= u <= x, 0 <= v <= y, u^2 + v^2 <= r^2} for x, y >= 0 (arrays)."""
    x = np.minimum(x, r)
    y = np.minimum(y, r)
    inside = x * x + y * y <= r * r
    u0 = np.sqrt(np.maximum(r * r - y * y, 0.0))           # the circle reaches height y at u0 <= x (when not inside)

    def prim(u):                                            # integral of sqrt(r^2 - u^2)
        return 0.5 * (u * np.sqrt(np.maximum(r * r - u * u, 0.0)) + r * r * np.arcsin(np.clip(u / r, -1.0, 1.0)))
    return np.where(inside, x * y, y * u0 + prim(x) - prim(u0))


def _signed_area(x, y, r):
    """Odd extension of ``_quadrant_area`` to all signs, so that rectangles follow by inclusion-exclusion."""
    return np.sign(x) * np.sign(y) * _quadrant_area(np.abs(x), np.abs(y), r)


def circle_pixel_overlap(dx0, dy0, r):
    """Exact area of the unit pixels [dx0, dx0+1] x [dy0, dy0+1] (coordinates relative to the circle centre) inside the
    circle of radius r."""
    x1, y1 = dx0 + 1.0, dy0 + 1.0
    return (_signed_area(x1, y1, r) - _signed_area(dx0, y1, r) - _signed_area(x1, dy0, r) + _signed_area(dx0, dy0, r))


def aperture_sums_exact(array, xx, yy, r):
    """Sum of ``array`` over circular apertures of radius r centred on (xx[i], yy[i]) (pixel centres at integer
    coordinates, as photutils), every pixel weighted by its exact overlap with the circle; the part of an aperture
    outside the frame contributes nothing.  All apertures in one vectorised pass (round 6: a ring at 128 px holds ~200
    apertures, and the S/N-scored grids call this per pixel of the test aperture: a Python loop over the apertures made a
    five-entry grid at 512 px cost 0.95 s of host time against 26 ms of device work)."""
    array = np.asarray(array, dtype=np.float64)
    ny, nx = array.shape
    xc = np.asarray(xx, dtype=np.float64).reshape(-1)
    yc = np.asarray(yy, dtype=np.float64).reshape(-1)
    if xc.size == 0:
        return np.zeros(0, dtype=np.float64)
    r = float(r)
    width = int(np.ceil(2.0 * r)) + 2                              # pixels an aperture's bounding box can span per axis
    off = np.arange(width)
    ix = np.floor(xc - r + 0.5).astype(np.int64)[:, None] + off    # (m, width) pixel columns / rows of every aperture's box
    iy = np.floor(yc - r + 0.5).astype(np.int64)[:, None] + off
    gx = ix - 0.5 - xc[:, None]                                    # lower-left pixel corners relative to the centre
    gy = iy - 0.5 - yc[:, None]
    w = circle_pixel_overlap(gx[:, None, :], gy[:, :, None], r)    # (m, rows, columns)
    inside = ((iy >= 0) & (iy < ny))[:, :, None] & ((ix >= 0) & (ix < nx))[:, None, :]
    vals = array[np.clip(iy, 0, ny - 1)[:, :, None], np.clip(ix, 0, nx - 1)[:, None, :]]
    # (photutils sums only the pixels the aperture touches: a NaN corner of the bounding box outside the circle must not
    #  poison the sum)
    covered = inside & (w > 0)
    return np.where(covered, w * np.where(covered, vals, 0.0), 0.0).sum(axis=(1, 2))


def _ring_geometry(array, source_xy, fwhm):
    """(cy, cx, separation, polar angle [rad], angular pitch [rad], number of apertures that fit) of the ring of 1-FWHM
    apertures through ``source_xy``: neighbours touch, i.e. the chord between their centres is one FWHM."""
    cy, cx = frame_center(array)
    dx, dy = float(source_xy[0]) - cx, float(source_xy[1]) - cy
    sep = np.hypot(dy, dx)
    if not sep > fwhm / 2:
        raise RuntimeError("`source_xy` is too close to the frame center")
    pitch = 2.0 * np.arcsin(0.5 * fwhm / sep)
    return cy, cx, sep, np.arctan2(dy, dx), pitch, int(np.floor(2 * np.pi / pitch))


def indep_ap_centers(array, source_xy, fwhm, exclude_negative_lobes=False, exclude_theta_range=None, no_gap=False):
    """Centres of the non-overlapping apertures at the separation of ``source_xy`` (reference snr_source.py:226-318); first
    entry = the test aperture.  Returns (yy, xx).

    Aperture m sits at polar angle phi0 - m * pitch (clockwise from the source, as the reference walks the ring), written
    here in closed form, sep * (cos, sin)(phi0 - m pitch), instead of the reference's chained rotations.  Dropped from the
    list: the two neighbours of the test aperture when ``exclude_negative_lobes``; every aperture whose angle in degrees
    falls inside ``exclude_theta_range`` -- compared, as the reference does, after the source angle has been lifted by
    whole turns above the range's upper end (and the lower end by one turn when the source itself lies inside)."""
    cy, cx, sep, phi0, pitch, count = _ring_geometry(array, source_xy, fwhm)
    if no_gap:
        count += 1
    m = np.arange(1, count)                                  # ring positions after the test aperture
    keep = np.ones(m.shape, dtype=bool)
    if exclude_negative_lobes:
        keep &= (m != 1) & (m != count - 1)
    if exclude_theta_range is not None:
        lo, hi = (float(v) for v in exclude_theta_range)
        deg0 = np.rad2deg(phi0)
        if lo < deg0 < hi:
            lo += 360.0
        if deg0 < hi:
            deg0 += 360.0 * np.ceil((hi - deg0) / 360.0)
            if deg0 < hi:                                    # (the reference adds turns WHILE below the upper end)
                deg0 += 360.0
        deg = deg0 - m * np.rad2deg(pitch)
        keep &= (deg < lo) | (deg > hi)
    phi = phi0 - m[keep] * pitch
    xs = np.concatenate(([float(source_xy[0]) - cx], sep * np.cos(phi))) + cx
    ys = np.concatenate(([float(source_xy[1]) - cy], sep * np.sin(phi))) + cy
    return ys, xs


def _small_sample_snr(test_flux, ring_fluxes):
    """[MAW14] eq. 9: (x1 - mean(x2)) / (s2 sqrt(1 + 1/n2)), s2 the sample standard deviation of the n2 ring apertures."""
    n2 = ring_fluxes.size
    spread = ring_fluxes.std(ddof=1)
    return (test_flux - ring_fluxes.mean()) / (spread * np.sqrt(1.0 + 1.0 / n2)), spread


def snr(array, source_xy, fwhm, full_output=False, array2=None, use2alone=False, exclude_negative_lobes=False,
        exclude_theta_range=None, plot=False, verbose=False):
    """Student-t S/N of [MAW14] (reference snr_source.py:321-456): flux of the 1-FWHM test aperture against the other
    apertures of its ring, with the small-sample penalty.  ``array2`` adds (or, with ``use2alone``, replaces) the noise
    apertures by those of a second frame at the same positions -- all of them, its test position included, as the
    reference does."""
    array = np.asarray(array)
    if array.ndim != 2:
        raise TypeError("Input array is not a frame or 2d array")
    if not isinstance(source_xy, tuple):
        raise TypeError("`source_xy` must be a tuple of floats")
    if array2 is not None and np.asarray(array2).shape != array.shape:
        raise TypeError("`array2` has not the same shape as input array")
    ys, xs = indep_ap_centers(array, source_xy, fwhm, exclude_negative_lobes, exclude_theta_range)
    sums = aperture_sums_exact(array, xs, ys, 0.5 * fwhm)
    test_flux, ring = float(sums[0]), sums[1:]
    if array2 is not None:
        other = aperture_sums_exact(array2, xs, ys, 0.5 * fwhm)
        ring = other if use2alone else np.concatenate((ring, other))
    value, spread = _small_sample_snr(test_flux, ring)
    if verbose:
        print("S/N at (x, y) = ({:.1f}, {:.1f}): {:.3f}   [test aperture {:.3f}; {} ring apertures: mean {:.3f}, std {:.3f}]".format(
            float(source_xy[0]), float(source_xy[1]), value, test_flux, ring.size, ring.mean(), spread))
    if full_output:
        return source_xy[1], source_xy[0], test_flux, ring, value
    return value


def disk_pixels(y, x, radius, shape=None):
    """(yy, xx) of ``skimage.draw.disk((y, x), radius)``: pixels with ((r-y)/R)^2 + ((c-x)/R)^2 < 1, row-major order."""
    r_lo, r_hi = int(np.ceil(y - radius)), int(np.floor(y + radius))
    c_lo, c_hi = int(np.ceil(x - radius)), int(np.floor(x + radius))
    rr, cc = np.mgrid[r_lo:r_hi + 1, c_lo:c_hi + 1]
    m = ((rr - y) / radius) ** 2 + ((cc - x) / radius) ** 2 < 1
    rr, cc = rr[m], cc[m]
    if shape is not None:
        ok = (rr >= 0) & (rr < shape[0]) & (cc >= 0) & (cc < shape[1])
        rr, cc = rr[ok], cc[ok]
    return rr, cc


def frame_report(array, fwhm, source_xy=None, verbose=True, **snr_arguments):
    """Per position: flux in the centred 1-FWHM aperture, S/N of the position itself, mean S/N over the pixels of that
    aperture (reference snr_source.py:515-590; returns (positions, fluxes, central S/N, mean S/N) as lists).  The automatic
    detection branch (``source_xy=None`` -> ``snrmap``) is outside the accelerated path."""
    array = np.asarray(array)
    if array.ndim != 2:
        raise TypeError("Array is not 2d.")
    if source_xy is None:
        raise NotImplementedError("frame_report without source_xy needs snrmap (outside the accelerated path)")
    if not isinstance(source_xy, (list, tuple)):
        raise TypeError("`source_xy` must be a tuple of floats or tuple of tuples")
    positions = source_xy if isinstance(source_xy[0], tuple) else [source_xy]
    table = []
    for px, py in positions:
        rows, cols = disk_pixels(py, px, 0.5 * fwhm)
        inside = np.array([snr(array, (float(c), float(r)), fwhm) for r, c in zip(rows, cols)])
        table.append((float(aperture_sums_exact(array, [px], [py], 0.5 * fwhm)[0]), snr(array, (px, py), fwhm), inside))
    if verbose:
        for (px, py), (flux, central, inside) in zip(positions, table):
            print("(x, y) = ({:.1f}, {:.1f}): flux in the 1xFWHM aperture {:.3f}, S/N of the position {:.3f}; over the {} "
                  "pixels of the aperture: mean S/N {:.3f}, max {:.3f}, std {:.3f}".format(
                      px, py, flux, central, inside.size, inside.mean(), inside.max(), inside.std(ddof=1)))
    return positions, [t[0] for t in table], [t[1] for t in table], [float(np.mean(t[2])) for t in table]


__all__ = ["snr", "indep_ap_centers", "frame_report", "aperture_sums_exact", "circle_pixel_overlap", "disk_pixels",
           "disk_mask"]
