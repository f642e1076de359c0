"""Where the cold path of one assembly goes: patch creation, first and second assembly, pattern,
D2H, scipy wrap -- wall clock around each phase of assemble.stiffness() at the chosen config.
usage: python tools/cold_path.py [c4|c3|c2] [--api]"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
t00 = time.perf_counter()
import pyiga_amd as iga
from pyiga_amd import _lib
t_import = time.perf_counter() - t00
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
dim, p, n = {'c4': (3, 4, 128), 'c3': (3, 2, 64), 'c2': (2, 3, 256)}[cfg]


def lap(msg, t0, ctx=None):
    if ctx is not None:
        ctx.sync()
    t1 = time.perf_counter()
    print('%-58s %9.2f ms' % (msg, 1e3 * (t1 - t0)))
    return t1


geo = iga.geometry.tensor_product(iga.geometry.line_segment(0.0, 1.0), iga.geometry.quarter_annulus()) if dim == 3 else iga.geometry.quarter_annulus()
print('import pyiga_amd %.1f ms' % (1e3 * t_import))
for rep in range(2):
    print('--- round', rep, '(0: cold process, 1: library and allocator warm)')
    t = time.perf_counter()
    kv = iga.bspline.make_knots(p, 0., 1., n)
    kvs = (kv,) * dim
    t = lap('knot vectors', t)
    patch = iga.assemblers.DevicePatch(kvs, geo)
    t = lap('DevicePatch (igx_patch_create: tables, plan, uploads)', t, patch.ctx)
    patch.assemble('stiffness', to_host=False)
    t = lap('first assembly (workspace allocation + kernels)', t, patch.ctx)
    print('      device time of the chain: %.2f ms' % patch.timing()['total_ms'])
    patch.assemble('stiffness', to_host=False)
    t = lap('second assembly', t, patch.ctx)
    if '--api' in sys.argv:
        ip, ix = patch.pattern()
        t = lap('pattern (k_pattern + D2H of indptr/indices)', t, patch.ctx)
        data = patch.assemble('stiffness', to_host=True)
        t = lap('assembly + D2H of the values', t, patch.ctx)
        import scipy.sparse
        A = scipy.sparse.csr_matrix((data, ix, ip), shape=patch.shape)
        t = lap('scipy csr_matrix wrap', t)
        del A, data, ip, ix
    patch.close()
    t = lap('patch.close()', t)
if '--api' in sys.argv:
    t = time.perf_counter()
    A = iga.assemble.stiffness(kvs, geo)
    lap('assemble.stiffness() end to end', t)
