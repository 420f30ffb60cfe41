"""linearcorex_amd: MI355X-native Linear CorEx fit path behind the reference's `Corex` API.

    from linearcorex_amd import Corex        # drop-in for `from linearcorex import Corex`
"""
from .corex import Corex, DeviceMoments, pick_n_hidden  # noqa: F401
from .preprocess import g, g_inv, mean_impute, random_impute  # noqa: F401

__version__ = "0.1.0"
