import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyiga_amd import bspline, geometry, assemblers
p, n = int(sys.argv[1]), int(sys.argv[2])
kv = bspline.make_knots(p, 0.0, 1.0, n)
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
asm = assemblers.ConvDiffAssembler3D((kv, kv, kv), geo, lambda x, y, z: 1.0 + x)
for _ in range(3):
    asm.patch.assemble('convdiff', to_host=False)
t = asm.patch.timing()
print('p=%d n=%d' % (p, n), {k: round(v, 3) for k, v in t.items() if k.endswith('_ms')})
