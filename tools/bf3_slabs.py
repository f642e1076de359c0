#!/usr/bin/env python3
"""Row slabs of a 3D stiffness patch against the whole patch, bit for bit, with the location of the first differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyiga_amd import bspline, geometry, assemblers, distributed

p, n, W = [int(x) for x in (sys.argv[1:4] + ['4', '40', '4'][len(sys.argv) - 1:])][:3]
kv = bspline.make_knots(p, 0.0, 1.0, n)
kvs = (kv, kv, kv)
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
N = kv.numdofs
full = assemblers.DevicePatch(kvs, geo)
data = full.assemble('stiffness', algo='sumfact', to_host=True).copy()
full.close()
import scipy.sparse
asm = assemblers.StiffnessAssembler3D(kvs, geo)
A = asm.assemble_csr()
indptr = A.indptr.astype(np.int64)
for r in range(W):
    lo, hi = distributed.slab_range(N, r, W, p)
    sl = assemblers.DevicePatch(kvs, geo, row0=(lo, hi))
    part = sl.assemble('stiffness', algo='sumfact', to_host=True)
    a, b = indptr[lo * N * N], (indptr[hi * N * N] if hi < N else data.size)
    ok = part.size == b - a and np.array_equal(part, data[a:b])
    print('slab %d rows [%d,%d): %s' % (r, lo, hi, 'identical' if ok else 'DIFFERENT'))
    if not ok and part.size == b - a:
        bad = np.nonzero(part != data[a:b])[0] + a
        rows = np.searchsorted(indptr, bad, side='right') - 1
        ur = np.unique(rows)
        print('   %d entries in %d rows; i0 %s i1 %s i2 %s' % (bad.size, ur.size, sorted(set((ur // (N * N)).tolist()))[:30], sorted(set(((ur // N) % N).tolist()))[:40], sorted(set((ur % N).tolist()))[:40]))
        for k in bad[:8]:
            rr = np.searchsorted(indptr, k, side='right') - 1
            col = A.indices[k]
            print('   row (%d,%d,%d) col (%d,%d,%d): slab %.6e full %.6e' % (rr // (N * N), (rr // N) % N, rr % N, col // (N * N), (col // N) % N, col % N, part[k - a], data[k]))
    sl.close()
