"""S/N of a test resolution element (host) and STIM detection maps (reference metrics/stim.py) from the derotated residual cube that ``pca`` leaves on the device."""
from .stim import stim_map, inverse_stim_map, normalized_stim_map  # noqa: F401
from .snr_source import snr, indep_ap_centers, frame_report  # noqa: F401
