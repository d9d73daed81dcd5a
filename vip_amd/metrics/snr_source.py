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
    outside the frame contributes nothing."""
    array = np.asarray(array, dtype=np.float64)
    ny, nx = array.shape
    out = np.zeros(len(xx), dtype=np.float64)
    for i, (xc, yc) in enumerate(zip(xx, yy)):
        x_lo, x_hi = int(np.floor(xc - r + 0.5)), int(np.ceil(xc + r - 0.5))
        y_lo, y_hi = int(np.floor(yc - r + 0.5)), int(np.ceil(yc + r - 0.5))
        xs = np.arange(max(x_lo, 0), min(x_hi, nx - 1) + 1)
        ys = np.arange(max(y_lo, 0), min(y_hi, ny - 1) + 1)
        if xs.size == 0 or ys.size == 0:
            continue
        gx, gy = np.meshgrid(xs - 0.5 - xc, ys - 0.5 - yc)            # lower-left pixel corners relative to the centre
        w = circle_pixel_overlap(gx, gy, float(r))
        sub = array[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1]
        covered = w > 0                      # (photutils sums only the pixels the aperture touches: a NaN corner of the
        out[i] = float(np.sum(w[covered] * sub[covered]))        # bounding box outside the circle must not poison the sum)
    return out


def indep_ap_centers(array, source_xy, fwhm, exclude_negative_lobes=False, exclude_theta_range=None, no_gap=False):
    """Centres of the non-overlapping apertures at the separation of ``source_xy`` (snr_source.py:226-318); first entry =
    the test aperture.  Returns (yy, xx)."""
    sourcex, sourcey = source_xy
    centery, centerx = frame_center(array)
    sep = np.sqrt((centery - float(sourcey)) ** 2 + (centerx - float(sourcex)) ** 2)
    theta_0 = np.rad2deg(np.arctan2(sourcey - centery, sourcex - centerx))
    if exclude_theta_range is not None:
        exc = list(exclude_theta_range)
    if not sep > (fwhm / 2):
        raise RuntimeError("`source_xy` is too close to the frame center")
    sign = -1                                               # clockwise, as the reference
    if exclude_theta_range is not None:
        if exc[0] < theta_0 < exc[1]:
            exc[0] += 360
        while theta_0 < exc[1]:
            theta_0 += 360
    theta = theta_0
    angle = np.arcsin(fwhm / 2.0 / sep) * 2
    number_apertures = int(np.floor(2 * np.pi / angle))
    if no_gap:
        number_apertures += 1
    yy, xx = [sourcey - centery], [sourcex - centerx]
    yy_all = np.zeros(number_apertures)
    xx_all = np.zeros(number_apertures)
    cosangle, sinangle = np.cos(angle), np.sin(angle)
    xx_all[0], yy_all[0] = sourcex - centerx, sourcey - centery
    for i in range(number_apertures - 1):
        xx_all[i + 1] = cosangle * xx_all[i] - sign * sinangle * yy_all[i]
        yy_all[i + 1] = cosangle * yy_all[i] + sign * sinangle * xx_all[i]
        theta += sign * np.rad2deg(angle)
        if exclude_negative_lobes and (i == 0 or i == number_apertures - 2):
            continue
        if exclude_theta_range is None or theta < exc[0] or theta > exc[1]:
            xx.append(cosangle * xx_all[i] - sign * sinangle * yy_all[i])
            yy.append(cosangle * yy_all[i] + sign * sinangle * xx_all[i])
    return np.array(yy) + centery, np.array(xx) + centerx


def snr(array, source_xy, fwhm, full_output=False, array2=None, use2alone=False, exclude_negative_lobes=False,
        exclude_theta_range=None, plot=False, verbose=False):
    """Student-t S/N of [MAW14] (snr_source.py:321-456): flux of the test aperture against the mean / sample standard
    deviation of the other apertures at the same separation, with the small-sample factor sqrt(1 + 1/n2)."""
    array = np.asarray(array)
    if array.ndim != 2:
        raise TypeError("Input array is not a frame or 2d array")
    if not isinstance(source_xy, tuple):
        raise TypeError("`source_xy` must be a tuple of floats")
    if array2 is not None and np.asarray(array2).shape != array.shape:
        raise TypeError("`array2` has not the same shape as input array")
    sourcex, sourcey = source_xy
    yy, xx = indep_ap_centers(array, source_xy, fwhm, exclude_negative_lobes, exclude_theta_range)
    rad = fwhm / 2.0
    fluxes = aperture_sums_exact(array, xx, yy, rad)
    if array2 is not None:
        fluxes2 = aperture_sums_exact(array2, xx, yy, rad)
        fluxes = np.concatenate(([fluxes[0]], fluxes2)) if use2alone else np.concatenate((fluxes, fluxes2))
    f_source = fluxes[0].copy()
    fluxes = fluxes[1:]
    n2 = fluxes.shape[0]
    backgr_apertures_std = fluxes.std(ddof=1)
    snr_vale = (f_source - fluxes.mean()) / (backgr_apertures_std * np.sqrt(1 + (1 / n2)))
    if verbose:
        print("S/N for the given pixel = {:.3f}".format(snr_vale))
        print("Integrated flux in FWHM test aperture = {:.3f}".format(f_source))
        print("Mean of background apertures integrated fluxes = {:.3f}".format(fluxes.mean()))
        print("Std-dev of background apertures integrated fluxes = {:.3f}".format(backgr_apertures_std))
    if full_output:
        return sourcey, sourcex, f_source, fluxes, snr_vale
    return snr_vale


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
    """Flux in a centred 1xFWHM aperture, S/N of the central pixel and mean S/N over the aperture's pixels for the given
    position(s) (snr_source.py:515-590).  The automatic detection branch (``source_xy=None`` -> ``snrmap``) is outside
    the accelerated path."""
    array = np.asarray(array)
    if array.ndim != 2:
        raise TypeError("Array is not 2d.")
    if source_xy is None:
        raise NotImplementedError("frame_report without source_xy needs snrmap (outside the accelerated path)")
    if isinstance(source_xy, (list, tuple)):
        if not isinstance(source_xy[0], tuple):
            source_xy = [source_xy]
    else:
        raise TypeError("`source_xy` must be a tuple of floats or tuple of tuples")
    obj_flux, meansnr_pixels, snr_centpx = [], [], []
    for x, y in source_xy:
        obj_flux_i = float(aperture_sums_exact(array, [x], [y], fwhm / 2.0)[0])
        yy, xx = disk_pixels(y, x, fwhm / 2)
        snr_pixels_i = [snr(array, (x_, y_), fwhm) for y_, x_ in zip(yy, xx)]
        meansnr_i = np.mean(snr_pixels_i)
        pxsnr_i = snr(array, (x, y), fwhm)
        obj_flux.append(obj_flux_i)
        meansnr_pixels.append(meansnr_i)
        snr_centpx.append(pxsnr_i)
        if verbose:
            print("Coords of chosen px (X,Y) = {:.1f}, {:.1f}".format(x, y))
            print("Flux in a centered 1xFWHM circular aperture = {:.3f}".format(obj_flux_i))
            print("Central pixel S/N = {:.3f}".format(pxsnr_i))
            print("Inside a centered 1xFWHM circular aperture:")
            print("Mean S/N (shifting the aperture center) = {:.3f}".format(meansnr_i))
            print("Max S/N (shifting the aperture center) = {:.3f}".format(np.max(snr_pixels_i)))
            print("stddev S/N (shifting the aperture center) = {:.3f}".format(np.std(snr_pixels_i, ddof=1)))
    return source_xy, obj_flux, snr_centpx, meansnr_pixels


__all__ = ["snr", "indep_ap_centers", "frame_report", "aperture_sums_exact", "circle_pixel_overlap", "disk_pixels",
           "disk_mask"]
