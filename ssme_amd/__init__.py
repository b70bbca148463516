"""ssme_amd: MI355X-native bootstrap-particle-filter core behind ssme's BSFilter<>::filter().

Python host mirror of the reference's filter interface over the C ABI (include/ssme_pf.h).
"""
from .filters import *  # noqa: F401,F403
from . import _capi  # noqa: F401

__version__ = "0.1.0"
