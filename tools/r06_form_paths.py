import sys, time, numpy as np
sys.path.insert(0, '.')
from pyiga_amd import bspline, geometry, assemble, assemblers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kv = bspline.make_knots(p, 0., 1., n)
geo = geometry.tensor_product(geometry.line_segment(0., 1.), geometry.quarter_annulus())
for form in ('(inner(grad(u), grad(v)) + u * v) * dx', '(2 * inner(grad(u), grad(v)) + 3 * u * v) * dx', '(inner(grad(u), grad(v)) + inner((x[1], -x[0], 1.0), grad(u)) * v) * dx',
             '((1 + x[0]) * inner(grad(u), grad(v)) + u * v) * dx'):
    asm = assemble.instantiate_assembler(form, (kv, kv, kv), {'geo': geo}, None) if hasattr(assemble, 'instantiate_assembler') else None
    t0 = time.perf_counter()
    patch = asm.patch
    for it in range(3):
        patch.assemble(getattr(asm, '_kind', 'form'), to_host=False)
        tm = patch.timing()
    print(form, '| path', sorted(patch.last_path()), '| total_ms %.3f' % tm['total_ms'], {k: round(v, 3) for k, v in tm.items() if k.endswith('_ms') and v > 0})
patch = assemblers.DevicePatch((kv, kv, kv), geo)
for it in range(3):
    patch.assemble('stiffness', to_host=False)
print('stiffness()', sorted(patch.last_path()), patch.timing()['total_ms'])
