"""Import shim for the *reference* package (vip_hci) -- TEST INFRASTRUCTURE ONLY.

This module exists so that ``oracle/gen_golden.py`` and ``oracle/check_vs_reference.py``
can import the read-only reference tree at ``/root/reference/src`` inside the build
container, where several of the reference's third-party dependencies (scikit-image,
astropy, photutils, opencv, ...) are not installed.  Missing top-level packages are
replaced by inert ``MagicMock`` modules, with real stand-ins only where the PSF-subtraction
hot path actually touches them:

* ``skimage.draw.disk``  (used by ``vip_hci.var.shapes.mask_circle``, shapes.py:88)
* ``astropy.utils.exceptions.AstropyWarning`` (warnings filter in derotation.py:33)
* ``astropy.stats.gaussian_fwhm_to_sigma / gaussian_sigma_to_fwhm`` (constants)

Nothing here is shipped to the GPU box in any useful form: ``/root/reference`` does not
exist there, and no product code, ``-m gpu`` test, ``smoke()`` or ``bench.py`` imports it.
"""
import importlib.abc
import importlib.machinery
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REFERENCE_SRC = "/root/reference/src"

MISSING = ('skimage', 'astropy', 'photutils', 'hciplot', 'emcee', 'nestle', 'corner',
           'dataclass_builder', 'cv2', 'ratelimit', 'requests')


def _disk(center, radius, shape=None):
    """Stand-in for ``skimage.draw.disk``: pixels with ((r-cy)/R)^2 + ((c-cx)/R)^2 < 1,
    clipped to ``shape``, in row-major order (the documented scikit-image rule)."""
    cy, cx = center
    r0, r1 = int(np.floor(cy - radius)), int(np.ceil(cy + radius)) + 1
    c0, c1 = int(np.floor(cx - radius)), int(np.ceil(cx + radius)) + 1
    if shape is not None:
        r0, c0, r1, c1 = max(r0, 0), max(c0, 0), min(r1, shape[0]), min(c1, shape[1])
    rr, cc = np.mgrid[r0:r1, c0:c1]
    m = ((rr - cy) / radius) ** 2 + ((cc - cx) / radius) ** 2 < 1
    return rr[m], cc[m]


class _Stub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__all__ = []
        return m

    def exec_module(self, module):
        if module.__name__ == 'astropy.utils.exceptions':
            module.AstropyWarning = type('AstropyWarning', (Warning,), {})
        if module.__name__ == 'astropy.stats':
            module.gaussian_fwhm_to_sigma = 0.42466090014400953
            module.gaussian_sigma_to_fwhm = 2.3548200450309493
        if module.__name__ == 'skimage.draw':
            module.disk = _disk


_installed = False


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_SRC, "vip_hci"))


def install():
    """Make ``import vip_hci`` resolve to the read-only reference tree."""
    global _installed, MISSING
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present (only exists in the build container)")
    # real packages that ARE installed must not be shadowed
    really_missing = []
    for name in MISSING:
        try:
            __import__(name)
        except Exception:
            really_missing.append(name)
    MISSING = tuple(really_missing)
    sys.meta_path.insert(0, _Stub())
    sys.path.insert(0, REFERENCE_SRC)
    _installed = True


def load():
    """Return a namespace with the reference functions on the hot path."""
    install()
    import types
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import vip_hci.psfsub as ps
        from vip_hci.psfsub.svd import svd_wrapper, get_eigenvectors, SVDecomposer
        from vip_hci.psfsub.pca_fullfr import _project_subtract
        from vip_hci.preproc import (cube_derotate, cube_collapse, frame_rotate,
                                     check_pa_vector)
        from vip_hci.preproc.derotation import (_find_indices_adi, _define_annuli,
                                                _compute_pa_thresh, rotate_fft)
        from vip_hci.preproc.cosmetics import frame_pad
        from vip_hci.metrics.stim import stim_map, inverse_stim_map, normalized_stim_map
        from vip_hci.var import (prepare_matrix, matrix_scaling, mask_circle,
                                 get_annulus_segments, frame_center, reshape_matrix)
    ns = types.SimpleNamespace(
        pca=ps.pca, pca_annular=ps.pca_annular, PCA_Params=ps.PCA_Params,
        PCA_ANNULAR_Params=ps.PCA_ANNULAR_Params,
        svd_wrapper=svd_wrapper, get_eigenvectors=get_eigenvectors,
        SVDecomposer=SVDecomposer, _project_subtract=_project_subtract,
        cube_derotate=cube_derotate, cube_collapse=cube_collapse,
        frame_rotate=frame_rotate, check_pa_vector=check_pa_vector,
        _find_indices_adi=_find_indices_adi, _define_annuli=_define_annuli,
        _compute_pa_thresh=_compute_pa_thresh, rotate_fft=rotate_fft,
        frame_pad=frame_pad, prepare_matrix=prepare_matrix,
        matrix_scaling=matrix_scaling, mask_circle=mask_circle,
        get_annulus_segments=get_annulus_segments, frame_center=frame_center,
        reshape_matrix=reshape_matrix, median_sub=ps.median_sub, stim_map=stim_map,
        inverse_stim_map=inverse_stim_map, normalized_stim_map=normalized_stim_map)
    return ns
