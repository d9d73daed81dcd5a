"""Drop-in for the PCA entry points of ``vip_hci.psfsub`` (reference psfsub/__init__.py:14-24)."""
from .pca_fullfr import pca, pca_many, PCA_Params  # noqa: F401
from .pca_local import pca_annular, PCA_ANNULAR_Params  # noqa: F401
from .svd import svd_wrapper, SVDecomposer, get_eigenvectors  # noqa: F401
from .medsub import median_sub, MEDIAN_SUB_Params  # noqa: F401
from .utils_pca import pca_grid, pca_annulus  # noqa: F401
