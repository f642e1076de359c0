import sys, time, numpy as np
sys.path.insert(0, '.')
from pyiga_amd import bspline, geometry, assemble
n, p = 128, 4
kv = bspline.make_knots(p, 0., 1., n)
geo = geometry.tensor_product(geometry.line_segment(0., 1.), geometry.quarter_annulus())
form = '(inner(grad(u), grad(v)) + inner((x[1], -x[0], 1.0), grad(u)) * v) * dx'
asm = assemble.instantiate_assembler(form, (kv, kv, kv), {'geo': geo}, None)
patch = asm.patch
for it in range(4):
    patch.assemble('form', to_host=False)
    tm = patch.timing()
print('nonsym form', sorted(patch.last_path()), {k: round(v, 3) for k, v in tm.items() if k.endswith('_ms') and v > 0})
