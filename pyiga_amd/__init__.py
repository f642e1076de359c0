"""pyiga_amd -- MI355X-native tensor-product IgA assembly.

Drop-in for the hot path of c-f-h/pyiga: ``assemble.stiffness()/mass()`` with a geometry map.
Python host code calls hand-written gfx950 HIP kernels through the C ABI in ``include/igx.h``
(``libigx.so``, loaded with ctypes).  Nothing here falls back to the CPU.

    from pyiga_amd import bspline, geometry, assemble
    kv = bspline.make_knots(3, 0.0, 1.0, 64)
    A = assemble.stiffness((kv, kv), geometry.quarter_annulus())      # scipy CSR
"""
__version__ = '0.1.0'

_max_threads = None


def get_max_threads():
    """pyiga/__init__.py:10-15.  The reference's thread pool has no counterpart here (the launch geometry of the
    kernels replaces it); the value is kept so that code written for pyiga keeps working."""
    global _max_threads
    if not _max_threads:
        import multiprocessing
        _max_threads = multiprocessing.cpu_count()
    return _max_threads


def set_max_threads(num):
    """pyiga/__init__.py:17-19 (no effect on the device path)."""
    global _max_threads
    _max_threads = num


from . import _lib            # noqa: F401
from . import bspline         # noqa: F401
from . import geometry        # noqa: F401
from . import quadrature      # noqa: F401
from . import assemblers      # noqa: F401
from . import assemble        # noqa: F401
from . import utils           # noqa: F401
from . import distributed     # noqa: F401
