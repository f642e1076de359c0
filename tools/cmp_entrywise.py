"""Default chain against the entry-wise kernels on one 3D patch, with the position of every mismatch (debugging aid).
usage: python tools/cmp_entrywise.py p n [geo]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import pyiga_amd as iga

p, n = int(sys.argv[1]), int(sys.argv[2])
kv = iga.bspline.make_knots(p, 0., 1., n)
geo = iga.geometry.tensor_product(iga.geometry.line_segment(0.0, 1.0), iga.geometry.quarter_annulus())
patch = iga.assemblers.DevicePatch((kv,) * 3, geo)
A = patch.csr('stiffness', algo='sumfact')
E = patch.csr('stiffness', algo='entrywise')
scale = abs(E).max()
D = abs(A - E).tocoo()
bad = ~(D.data <= 1e-12 * scale)
print('p', p, 'n', n, 'nnz', E.nnz, 'bad', int(bad.sum()), 'max rel diff', D.data.max() / scale if D.nnz else 0.0, 'path', patch.last_path())
if bad.sum():
    N = n + p
    r, c = D.row[bad], D.col[bad]
    i0, i1, i2 = r // (N * N), (r // N) % N, r % N
    j0, j1, j2 = c // (N * N), (c // N) % N, c % N
    print('lower-triangle bad:', int((c <= r).sum()), 'upper:', int((c > r).sum()))
    low = c <= r
    for name, v in (('i0', i0[low]), ('i1', i1[low]), ('i2', i2[low]), ('j0-i0', (j0 - i0)[low]), ('j1-i1', (j1 - i1)[low]), ('j2-i2', (j2 - i2)[low])):
        print('  ', name, sorted(set(v.tolist()))[:60])
    rel = (D.data[bad] / scale)
    print('   rel diffs: min %.3e median %.3e max %.3e' % (rel.min(), np.median(rel), rel.max()))
