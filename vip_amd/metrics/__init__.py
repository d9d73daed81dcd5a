"""STIM detection maps (reference metrics/stim.py) from the derotated residual cube that ``pca`` leaves on the device."""
from .stim import stim_map, inverse_stim_map, normalized_stim_map  # noqa: F401
