"""Race hunt: repeat an assembly many times and require bit-identical values every time (the final stage waits on
its LDS-DMA queue with counted vmcnt; an under-wait would show up as a run-to-run difference)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyiga_amd import bspline, geometry, assemblers
geo3 = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
cases = [('3D p=2 n=40', (bspline.make_knots(2, 0., 1., 40),) * 3, geo3, 'stiffness', 150),
         ('3D p=4 n=20', (bspline.make_knots(4, 0., 1., 20),) * 3, geo3, 'stiffness', 150),
         ('3D p=5 n=13', (bspline.make_knots(5, 0., 1., 13),) * 3, geo3, 'mass', 100),
         ('3D p=3 n=17,9,31', (bspline.make_knots(3, 0., 1., 17), bspline.make_knots(3, 0., 1., 9), bspline.make_knots(3, 0., 1., 31)), geo3, 'stiffness', 150),
         ('2D p=3 n=256', (bspline.make_knots(3, 0., 1., 256),) * 2, geometry.quarter_annulus(), 'stiffness', 200),
         ('3D p=4 n=128', (bspline.make_knots(4, 0., 1., 128),) * 3, geo3, 'stiffness', 12)]
bad = 0
for name, kvs, geo, kind, reps in cases:
    patch = assemblers.DevicePatch(kvs, geo)
    ref = None
    for r in range(reps):
        d = patch.assemble(kind, algo='sumfact', to_host=True)
        h = hashlib.sha1(d.tobytes()).hexdigest()
        if ref is None:
            ref = h
        elif h != ref:
            bad += 1
            print('MISMATCH', name, 'rep', r)
            break
    print(name, 'reps', reps, 'nnz', patch.nnz, 'ok' if h == ref else 'FAILED', flush=True)
    patch.close()
conv = assemblers.ConvDiffAssembler3D((bspline.make_knots(3, 0., 1., 24),) * 3, geo3, lambda x, y, z: 1.0 + x)
ref = None
for r in range(100):
    h = hashlib.sha1(conv.patch.assemble('convdiff', algo='sumfact', to_host=True).tobytes()).hexdigest()
    ref = ref or h
    if h != ref:
        bad += 1
        print('MISMATCH convdiff rep', r)
        break
print('convdiff p=3 n=24 reps 100', 'ok' if h == ref else 'FAILED')
sys.exit(1 if bad else 0)
