"""String enums of the hot-path switches (values identical to the reference's
config/paramenum.py:8-178, so ``'lapack' == SvdMode.LAPACK``)."""
from enum import Enum

ALGO_KEY = "algo_params"


class SvdMode(str, Enum):
    LAPACK = "lapack"
    ARPACK = "arpack"
    EIGEN = "eigen"
    RANDSVD = "randsvd"
    CUPY = "cupy"
    EIGENCUPY = "eigencupy"
    RANDCUPY = "randcupy"
    PYTORCH = "pytorch"
    EIGENPYTORCH = "eigenpytorch"
    RANDPYTORCH = "randpytorch"


class Scaling(str, Enum):
    TEMPMEAN = "temp-mean"
    SPATMEAN = "spat-mean"
    TEMPSTANDARD = "temp-standard"
    SPATSTANDARD = "spat-standard"


class Adimsdi(str, Enum):
    DOUBLE = "double"
    SINGLE = "single"


class Imlib(str, Enum):
    OPENCV = "opencv"
    SKIMAGE = "skimage"
    NDIMAGE = "ndimage"
    VIPFFT = "vip-fft"


class Interpolation(str, Enum):
    NEARNEIG = "nearneig"
    BILINEAR = "bilinear"
    BIQUADRATIC = "biquadratic"
    BICUBIC = "bicubic"
    BIQUARTIC = "biquartic"
    BIQUINTIC = "biquintic"
    LANCZOS4 = "lanczos4"


class Collapse(str, Enum):
    MEDIAN = "median"
    MEAN = "mean"
    SUM = "sum"
    TRIMMEAN = "trimmean"


__all__ = ["ALGO_KEY", "SvdMode", "Scaling", "Adimsdi", "Imlib", "Interpolation", "Collapse"]
