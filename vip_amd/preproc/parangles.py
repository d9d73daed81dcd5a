"""preproc/parangles.py:405-458 of the reference (host)."""
import numpy as np


def check_pa_vector(angle_list, unit="deg"):
    """Degrees, non-negative, and no jump of more than 180 deg between consecutive values."""
    if unit not in ("deg", "rad"):
        raise ValueError("The input unit should either be 'deg' or 'rad'")
    angle_list = np.array(angle_list, dtype=float, copy=True)
    if unit == "rad":
        angle_list = np.rad2deg(angle_list)
    angle_list[angle_list < 0] += 360
    if angle_list.size > 1 and np.any(np.abs(np.diff(angle_list)) > 180):
        angle_list[angle_list < 180] += 360
    return angle_list
