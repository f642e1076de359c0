import sys, os, time
sys.path.insert(0, '.')
import numpy as np
import pyiga_amd as iga
from pyiga_amd import bspline, geometry, assemble, assemblers
g = geometry
def rel(A, B):
    D = abs(A - B); return (D.max() if D.nnz else 0.0) / abs(B).max()
cases = [((3, 3, 3), (5, 6, 7), g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus())),
         ((4, 2, 3), (4, 7, 5), g.tensor_product(g.line_segment(0.5, 2.0, intervals=3), g.bspline_quarter_annulus())),
         ((2, 2, 2), (6, 6, 6), g.unit_cube()),
         ((1, 3, 2), (5, 4, 6), g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus()))]
for ps, ns, geo in cases:
    kvs = tuple(bspline.make_knots(p, 0.0, 1.0, n) for p, n in zip(ps, ns))
    assert geometry.split_axis0(geo) is not None
    for kind in ('mass', 'stiffness'):
        os.environ['IGX_SEPARABLE'] = '1'
        A = getattr(assemble, kind)(kvs, geo)
        del os.environ['IGX_SEPARABLE']
        B = getattr(assemble, kind)(kvs, geo)
        print(ps, ns, kind, 'rel', rel(A, B), 'sym', abs(A - A.T).max(), np.array_equal(A.indices, B.indices))
print('not separable:', geometry.split_axis0(g.twisted_box()), geometry.split_axis0(g.tensor_product(g.quarter_annulus(), g.line_segment(0.0, 1.0))) is not None)
