"""kwargs plumbing of the reference's public entry points (config/utils_param.py:61-164)."""
from functools import lru_cache
from inspect import signature

KWARGS_EXCEPTIONS = []


def separate_kwargs_dict(initial_kwargs, parent_class):
    """Split kwargs into the fields of ``parent_class`` and the rest (``rot_options``)."""
    class_params, more_params = {}, {}
    for key, value in initial_kwargs.items():
        if hasattr(parent_class, key) or key in KWARGS_EXCEPTIONS:
            class_params[key] = value
        else:
            more_params[key] = value
    return class_params, more_params


@lru_cache(maxsize=None)
def _parameter_names(fkt):
    # (inspect.signature costs 50 us per call: a tenth of the host time of a full-frame pca() front)
    return tuple(signature(fkt).parameters)


def setup_parameters(params_obj, fkt, **add_params):
    """Pick from ``params_obj`` (+ ``add_params``, which win) the arguments ``fkt`` accepts."""
    wanted = _parameter_names(fkt)
    allp = dict(vars(params_obj))
    allp.update(add_params)
    return {k: allp[k] for k in wanted if k in allp}
