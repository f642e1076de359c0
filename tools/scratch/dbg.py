import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from pyiga_amd import bspline, geometry, assemblers, utils
kv = bspline.make_knots(2, 0., 1., 10)
geo = geometry.twisted_box()
R = utils.read_sparse_matrix('tests/golden/poisson_neu_d3_p2_n10_stiff.mtx.gz')
for sel in ('valu', 'mfma'):
    os.environ['IGX_FINAL'] = sel
    A = assemblers.DevicePatch((kv,)*3, geo).csr('stiffness', algo='sumfact')
    print(sel, 'vs fixture', abs(A - R).max(), 'sym', abs(A - A.T).max())
E = assemblers.DevicePatch((kv,)*3, geo).csr('stiffness', algo='entrywise')
print('entrywise vs fixture', abs(E - R).max())
