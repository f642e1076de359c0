#!/usr/bin/env python3
"""Degrees 6 and 7 on the sum-factorised stages against the entry-wise kernels (and timing of both)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyiga_amd as iga
from pyiga_amd import bspline, geometry, assemblers

def rel(A, B):
    return abs(A - B).max() / abs(B).max()

cyl = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
ann = geometry.quarter_annulus()
for p in (5, 6, 7):
    for dim, n in ((2, 24), (3, 7)):
        kvs = tuple(bspline.make_knots(p, 0.0, 1.0, n + k) for k in range(dim))
        geo = ann if dim == 2 else cyl
        for kind, cls in (('mass', assemblers.MassAssembler2D if dim == 2 else assemblers.MassAssembler3D),
                          ('stiffness', assemblers.StiffnessAssembler2D if dim == 2 else assemblers.StiffnessAssembler3D)):
            asm = cls(kvs, geo)
            try:
                t0 = time.perf_counter(); A = asm.assemble_csr(algo='sumfact'); t1 = time.perf_counter()
                ms_sf = asm.patch.timing()['total_ms']
            except Exception as e:
                print(p, dim, kind, 'sumfact FAILED:', str(e)[:150]); continue
            B = asm.assemble_csr(algo='entrywise'); ms_en = asm.patch.timing()['total_ms']
            print('p=%d %dD %-9s sumfact %.3f ms  entrywise %.3f ms  rel diff %.2e  path %s' % (p, dim, kind, ms_sf, ms_en, rel(A, B), asm.patch.last_path()))
        if dim == 3:
            dc = lambda x, y, z: 1.0 + x * y
            asm = assemblers.ConvDiffAssembler3D(kvs, geo, dc)
            try:
                A = asm.assemble_csr(algo='sumfact'); ms_sf = asm.patch.timing()['total_ms']
                B = asm.assemble_csr(algo='entrywise'); ms_en = asm.patch.timing()['total_ms']
                print('p=%d 3D convdiff  sumfact %.3f ms  entrywise %.3f ms  rel diff %.2e' % (p, ms_sf, ms_en, rel(A, B)))
            except Exception as e:
                print(p, 'convdiff FAILED', str(e)[:150])
    # unequal degrees incl. a high one
    kvs = (bspline.make_knots(p, 0.0, 1.0, 5), bspline.make_knots(2, 0.0, 1.0, 9, mult=2), bspline.make_knots(3, 0.0, 1.0, 6))
    asm = assemblers.StiffnessAssembler3D(kvs, cyl)
    try:
        A = asm.assemble_csr(algo='sumfact'); B = asm.assemble_csr(algo='entrywise')
        print('p=(%d,2,3) stiffness rel diff %.2e' % (p, rel(A, B)))
    except Exception as e:
        print(p, 'mixed FAILED', str(e)[:150])
