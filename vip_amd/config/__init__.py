from .paramenum import *        # noqa: F401,F403
from .utils_param import separate_kwargs_dict, setup_parameters  # noqa: F401
