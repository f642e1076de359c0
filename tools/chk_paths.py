import os, sys, numpy as np
sys.path.insert(0, '.')
import pyiga_amd
from pyiga_amd import bspline, geometry, assemblers
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
for path in ('fused', 'unfused'):
    os.environ['IGX_PATH'] = path
    kv = bspline.make_knots(4, 0., 1., 16)
    P = assemblers.DevicePatch((kv,)*3, geo)
    A = P.csr('stiffness', algo='sumfact')
    print(path, P.timing())
