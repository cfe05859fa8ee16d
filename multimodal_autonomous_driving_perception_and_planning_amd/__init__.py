"""avhot: MI355X-native hot path of the perception -> tracking -> planning loop.

Drop-in class surfaces (same names, arguments and error behaviour as the reference's
src.perception / src.tracking / src.state_estimation / src.planning packages) over
hand-written HIP kernels in libavhot.so (include/avhot.h).  No CPU fallback exists.
"""
__version__ = "0.1.0"

from . import _native  # noqa: F401  (does not load the library until first use)
