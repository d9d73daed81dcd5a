"""var/coords.py:27-100 of the reference (host-side geometry)."""
import numpy as np


def frame_center(array, verbose=False):
    """(cy, cx): dim/2 for even sizes, dim/2 - 0.5 for odd ones, as ints (i.e. dim // 2)."""
    if array.ndim not in (2, 3, 4):
        raise ValueError("`array` is not a 2d, 3d or 4d array")
    shape = array.shape[-2:]
    cy = shape[0] / 2 - (0.5 if shape[0] % 2 else 0)
    cx = shape[1] / 2 - (0.5 if shape[1] % 2 else 0)
    if verbose:
        print("Center px coordinates at x,y = ({}, {})".format(cx, cy))
    return int(cy), int(cx)


def dist(yc, xc, y1, x1):
    return np.sqrt(np.power(yc - y1, 2) + np.power(xc - x1, 2))
