#!/usr/bin/env python3
"""Where does assemble.stiffness(kvs, geo) at C4 spend its second?  (patch set-up, pattern + its copy, device assembly, copy of the
values, scipy wrapper)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse
from pyiga_amd import bspline, geometry, assemblers, assemble

kv = bspline.make_knots(4, 0.0, 1.0, 128)
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
for rep in range(2):
    t = [time.perf_counter()]
    patch = assemblers.DevicePatch((kv, kv, kv), geo); t.append(time.perf_counter())
    indptr, indices = patch.pattern(); t.append(time.perf_counter())
    data = patch.assemble('stiffness', to_host=True); t.append(time.perf_counter())
    A = scipy.sparse.csr_matrix((data, indices, indptr), shape=patch.shape); t.append(time.perf_counter())
    patch.close(); t.append(time.perf_counter())
    names = ['patch', 'pattern (device + 6.4 GB to host)', 'assemble (device + 12.75 GB to host)', 'scipy csr_matrix', 'patch.close']
    print('rep %d: ' % rep + ', '.join('%s %.3f s' % (n, b - a) for n, a, b in zip(names, t[:-1], t[1:])), flush=True)
    del A, data, indptr, indices
    t0 = time.perf_counter(); A = assemble.stiffness((kv, kv, kv), geo); print('  assemble.stiffness(): %.3f s' % (time.perf_counter() - t0)); del A
