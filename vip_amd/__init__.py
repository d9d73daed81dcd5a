"""vip_amd -- MI355X-native drop-in for the ADI PSF-subtraction hot path of VIP (vip_hci).

Mirrors the reference's module layout for that path only:

    vip_amd.psfsub   : pca, pca_annular, median_sub, PCA_Params, PCA_ANNULAR_Params, svd.svd_wrapper
    vip_amd.preproc  : cube_derotate, frame_rotate, cube_collapse, check_pa_vector
    vip_amd.metrics  : stim_map, inverse_stim_map, normalized_stim_map
    vip_amd.var      : prepare_matrix, matrix_scaling, mask_circle, get_annulus_segments, frame_center, ...

All arithmetic runs in hand-written HIP kernels (vip_amd/csrc -> libvipmi.so, C ABI in
include/vipmi.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"

__all__ = ["psfsub", "preproc", "var", "config", "backend", "dist", "metrics"]


def __getattr__(name):
    if name in __all__:
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
