"""ctypes binding of libvipmi.so (the C ABI declared in include/vipmi.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C vip_amd/csrc``.  There is NO
CPU fallback: if the library is missing, or no MI355X is visible, every compute entry point raises.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIPMI_LIB_PATH") or os.path.join(_HERE, "libvipmi.so")   # (override: A/B builds)

c_f32p = ctypes.c_void_p
i64 = ctypes.c_int64

# name -> (argtypes after ctx, has_ctx)
_SIGS = {
    "vipmi_version": ([], False, ctypes.c_int),
    "vipmi_last_error": ([], False, ctypes.c_char_p),
    "vipmi_create": ([ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)], False, ctypes.c_int),
    "vipmi_destroy": ([], True, ctypes.c_int),
    "vipmi_trim": ([], True, ctypes.c_int),
    "vipmi_set_stream": ([ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_synchronize": ([], True, ctypes.c_int),
    "vipmi_check_deferred": ([], True, ctypes.c_int),
    "vipmi_gate_create": ([ctypes.POINTER(ctypes.c_void_p)], False, ctypes.c_int),
    "vipmi_gate_destroy": ([ctypes.c_void_p], False, ctypes.c_int),
    "vipmi_set_gate": ([ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_set_option": ([ctypes.c_char_p, i64], True, ctypes.c_int),
    "vipmi_get_option": ([ctypes.c_char_p], True, i64),
    "vipmi_stage_ms": ([ctypes.c_char_p], True, ctypes.c_float),
    "vipmi_stage_count": ([ctypes.c_char_p], True, ctypes.c_int),
    "vipmi_reset_timers": ([], True, ctypes.c_int),
    "vipmi_scale_f32": ([c_f32p, c_f32p, i64, i64, ctypes.c_int], True, ctypes.c_int),
    "vipmi_apply_mask_f32": ([c_f32p, c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_float], True, ctypes.c_int),
    "vipmi_gram_f32": ([c_f32p, i64, i64, i64, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_gram_batched_f32": ([c_f32p, i64, i64, i64, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_cross_gram_f32": ([c_f32p, i64, c_f32p, i64, i64, i64, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_eigh_f64": ([ctypes.c_void_p, i64, i64, ctypes.c_void_p, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_eigh_spectrum_f64": ([ctypes.c_void_p, i64, i64, i64, ctypes.c_void_p, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_eigh_topk_f64": ([ctypes.c_void_p, i64, i64, i64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p], True,
                            ctypes.c_int),
    "vipmi_eigh_topk_fast_f64": ([ctypes.c_void_p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)],
                                 True, ctypes.c_int),
    "vipmi_rowspace_gemm_f32": ([c_f32p, c_f32p, i64, i64, i64, c_f32p, c_f32p], True, ctypes.c_int),
    "vipmi_subtract_gemm_f32": ([c_f32p, c_f32p, c_f32p, i64, i64, i64, c_f32p, c_f32p], True, ctypes.c_int),
    "vipmi_lincomb_f32": ([c_f32p, c_f32p, ctypes.c_float, ctypes.c_float, i64, c_f32p], True, ctypes.c_int),
    "vipmi_zoom_frames_f32": ([c_f32p, i64, i64, c_f32p, c_f32p, ctypes.c_void_p, i64, i64, c_f32p, c_f32p], True,
                              ctypes.c_int),
    "vipmi_derotate_f32": ([c_f32p, ctypes.c_void_p, i64, i64, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int],
                           True, ctypes.c_int),
    "vipmi_derotate_maskval_f32": ([c_f32p, ctypes.c_void_p, i64, i64, c_f32p, ctypes.c_float, ctypes.c_int], True,
                                   ctypes.c_int),
    "vipmi_rotate_interp_f32": ([c_f32p, ctypes.c_void_p, i64, i64, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                 ctypes.c_int, c_f32p],
                                True, ctypes.c_int),
    "vipmi_collapse_f32": ([c_f32p, i64, i64, ctypes.c_int, c_f32p, i64, c_f32p], True, ctypes.c_int),
    "vipmi_project_batched_f32": ([c_f32p, c_f32p, i64, i64, i64, i64, c_f32p], True, ctypes.c_int),
    "vipmi_collapse_batched_f32": ([c_f32p, i64, i64, i64, ctypes.c_int, c_f32p, i64, c_f32p], True, ctypes.c_int),
    "vipmi_gather_f32": ([c_f32p, i64, i64, ctypes.c_void_p, i64, c_f32p], True, ctypes.c_int),
    "vipmi_scatter_f32": ([c_f32p, i64, i64, ctypes.c_void_p, i64, c_f32p], True, ctypes.c_int),
    "vipmi_subset_median_sub_f32": ([c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, c_f32p], True, ctypes.c_int),
    "vipmi_annular_residuals_multi_f32": ([c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, ctypes.c_void_p, i64,
                                           c_f32p], True, ctypes.c_int),
    "vipmi_annular_residuals_f32": ([c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, i64, c_f32p],
                                    True, ctypes.c_int),
    "vipmi_annular_subgrams_f64": ([c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, i64, ctypes.c_void_p,
                                    ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_center_f64": ([ctypes.c_void_p, i64, i64, ctypes.c_int, c_f32p, ctypes.c_void_p, c_f32p], True, ctypes.c_int),
    "vipmi_gram_offset_f64": ([c_f32p, ctypes.c_void_p, i64, i64, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_spat_center_f64": ([ctypes.c_void_p, i64, i64, i64, ctypes.c_int, c_f32p, ctypes.c_void_p, c_f32p, ctypes.c_void_p], True,
                              ctypes.c_int),
    "vipmi_gram_offset_u_f64": ([c_f32p, ctypes.c_void_p, ctypes.c_void_p, i64, i64, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_annular_apply_mu_u_f32": ([c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, i64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64, c_f32p, ctypes.c_void_p, c_f32p], True,
                                     ctypes.c_int),
    "vipmi_annular_apply_mu_f32": ([c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, i64, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64, c_f32p, c_f32p], True, ctypes.c_int),
    "vipmi_annular_eigh_f64": ([ctypes.c_void_p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, i64, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_annular_gram_all_f32": ([c_f32p, i64, i64, ctypes.c_void_p, i64, i64, ctypes.c_void_p, i64, c_f32p, ctypes.c_void_p],
                                   True, ctypes.c_int),
    "vipmi_annular_apply_all_f32": ([c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, ctypes.c_void_p, ctypes.c_void_p, i64,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64, i64, c_f32p, c_f32p],
                                    True, ctypes.c_int),
    "vipmi_annular_gram_all_f64": ([ctypes.c_void_p, i64, i64, ctypes.c_void_p, i64, i64, ctypes.c_void_p, i64, ctypes.c_int, c_f32p,
                                    ctypes.c_void_p, c_f32p, ctypes.c_void_p], True, ctypes.c_int),
    "vipmi_annular_apply_f32": ([c_f32p, i64, i64, ctypes.c_void_p, ctypes.c_void_p, i64, i64, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64, c_f32p], True, ctypes.c_int),
    "vipmi_pca_project_f32": ([c_f32p, i64, c_f32p, i64, i64, i64, c_f32p, c_f32p, c_f32p, ctypes.c_void_p],
                              True, ctypes.c_int),
    "vipmi_pca_4d_f32": ([c_f32p, ctypes.c_void_p, i64, i64, i64, i64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                          c_f32p, c_f32p], True, ctypes.c_int),
    "vipmi_pca_fullframe_f64": ([ctypes.c_void_p, ctypes.c_void_p, i64, i64, i64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                 c_f32p, c_f32p, c_f32p, c_f32p, c_f32p], True, ctypes.c_int),
    "vipmi_pca_fullframe_f32": ([c_f32p, ctypes.c_void_p, i64, i64, i64, ctypes.c_int, ctypes.c_void_p,
                                 ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p], True, ctypes.c_int),
    "vipmi_pca_fullframe_hostin_f32": ([ctypes.c_void_p, c_f32p, ctypes.c_void_p, i64, i64, i64, ctypes.c_void_p, ctypes.c_int, c_f32p,
                                        c_f32p, c_f32p, c_f32p, c_f32p], True, ctypes.c_int),
    "vipmi_rccl_load": ([ctypes.c_char_p], False, ctypes.c_int),
    "vipmi_rccl_unique_id": ([ctypes.c_void_p], False, ctypes.c_int),
    "vipmi_rccl_comm_create": ([ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)], True,
                               ctypes.c_int),
    "vipmi_rccl_comm_destroy": ([ctypes.c_void_p], False, ctypes.c_int),
    "vipmi_pca_fullframe_sharded_f32": ([ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p, i64, i64, i64,
                                         ctypes.c_int, c_f32p], True, ctypes.c_int),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None
_lock = threading.Lock()


class VipmiError(RuntimeError):
    pass


def load():
    """Load libvipmi.so (no GPU needed to load; needed to create a context)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise VipmiError(
                "libvipmi.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C vip_amd/csrc`. vip_amd has no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (args, has_ctx, res) in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = ([ctypes.c_void_p] if has_ctx else []) + list(args)
            fn.restype = res
        _lib = lib
        return _lib


def last_error():
    return load().vipmi_last_error().decode("utf-8", "replace")


# status -> Python exception type (mirrors the reference's error conventions)
def raise_for_status(status, what=""):
    if status == 0:
        return
    msg = last_error() or what
    if status == -1:
        raise ValueError(msg)
    if status == -5:
        raise NotImplementedError(msg)
    if status == -3:
        raise MemoryError(msg)
    raise VipmiError("%s (vipmi status %d)" % (msg, status))
