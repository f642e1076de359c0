"""Row sums of the stiffness matrix (K 1 = 0) at a size beyond the entry-wise kernel: which rows are off (debugging aid).
usage: python tools/rowsum_check.py p n [repeats]"""
import sys
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import pyiga_amd as iga

p, n = int(sys.argv[1]), int(sys.argv[2])
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 2
kv = iga.bspline.make_knots(p, 0., 1., n)
kvs = (kv,) * 3
geo = iga.geometry.tensor_product(iga.geometry.line_segment(0.0, 1.0), iga.geometry.quarter_annulus())
asm = iga.assemblers.StiffnessAssembler3D(kvs, geo)
N = n + p
ntot = N ** 3
# CSR row pointer of a tensor-product pattern
c = np.minimum(np.arange(N) + p, N - 1) - np.maximum(np.arange(N) - p, 0) + 1
cnt = (c[:, None, None] * c[None, :, None] * c[None, None, :]).reshape(-1)
indptr = np.concatenate(([0], np.cumsum(cnt)))
prev = None
for it in range(rep):
    data = asm.patch.assemble('stiffness', algo='sumfact', to_host=True)
    scale = np.abs(data).max()
    rowsum = np.add.reduceat(data, indptr[:-1])
    bad = np.nonzero(~(np.abs(rowsum) <= 1e-10 * scale))[0]
    print('run', it, 'nan', int(np.isnan(data).sum()), 'bad rows', bad.size, 'max |rowsum|/scale', np.nanmax(np.abs(rowsum)) / scale)
    if bad.size:
        i0, i1, i2 = bad // (N * N), (bad // N) % N, bad % N
        print('   i0:', sorted(set(i0.tolist()))[:40])
        print('   i1:', sorted(set(i1.tolist()))[:40])
        print('   i2:', sorted(set(i2.tolist()))[:60])
        if prev is not None: print('   same rows as previous run:', np.array_equal(prev, bad))
    prev = bad
