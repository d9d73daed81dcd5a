from .parangles import check_pa_vector  # noqa: F401
from .derotation import (cube_derotate, frame_rotate, _find_indices_adi, _compute_pa_thresh,  # noqa: F401
                         _define_annuli)
from .subsampling import cube_collapse  # noqa: F401
