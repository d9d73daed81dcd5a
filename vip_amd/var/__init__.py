from .coords import frame_center, dist  # noqa: F401
from .shapes import (mask_circle, get_annulus_segments, prepare_matrix, matrix_scaling,  # noqa: F401
                     reshape_matrix, disk_mask)
