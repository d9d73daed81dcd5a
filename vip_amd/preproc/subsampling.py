"""cube_collapse (preproc/subsampling.py:30-116) on the device."""
import numpy as np

from .. import backend as B


def cube_collapse(cube, mode="median", n=50, w=None):
    """Collapse a 3-D cube along axis 0 (4-D: axis 1).  NaN-aware like the reference's
    nanmedian/nanmean/nansum/nanmax.  numpy in -> numpy out (float32 in -> float32 out),
    cuda tensor in -> cuda tensor out."""
    if cube.ndim not in (3, 4):
        raise TypeError("The input array is not a cube or 3d array.")
    mode = str(getattr(mode, "value", mode))
    if mode == "wmean":
        if w is None:
            raise ValueError("Weights have to be provided for weighted mean mode")
        if len(w) != cube.shape[0]:
            raise TypeError("Weights need same length as cube")
    if mode not in B.COLLAPSE_MODES or mode == "stim":
        raise TypeError("mode not recognized")
    dev_in = B.is_device_tensor(cube)
    t = B.to_device_f32(cube)
    if cube.ndim == 3:
        out = B.collapse(t, mode, w=w, trim_n=n)
    else:
        torch = B._torch()
        out = torch.stack([B.collapse(t[j], mode, w=w, trim_n=n) for j in range(t.shape[0])])
    if dev_in:
        return out
    res = out.cpu().numpy()
    if cube.dtype == np.float64:
        res = res.astype(np.float64)
    return res
