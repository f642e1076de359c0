#!/usr/bin/env python3
"""Randomised cross-check on the GPU box of the convection-diffusion chain (k_geoA non-symmetric + k_bf2, or the stage
kernels where the library chooses them): sum-factorised against entry-wise kernels, full patch and row slabs, over random
degrees, sizes, geometries and coefficients (sampled / affine).  usage: python3 tools/fuzz_convdiff.py [ncases] [seed]"""
import sys

import numpy as np
import scipy.sparse

sys.path.insert(0, '.')
import pyiga_amd as iga


def rel(A, B):
    D = abs(A - B)
    return (D.max() if D.nnz else 0.0) / abs(B).max()


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    g = iga.geometry
    geos = [lambda: g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus()), g.twisted_box, lambda: g.unit_cube(),
            lambda: g.tensor_product(g.line_segment(0.0, 2.0, intervals=3), g.bspline_quarter_annulus()),
            lambda: g.tensor_product(g.quarter_annulus(), g.line_segment(0.0, 1.0)),        # degree 2 (NURBS) along axis 0
            lambda: g.tensor_product(g.bspline_quarter_annulus(), g.line_segment(0.0, 1.5, intervals=2))]
    worst = 0.0
    for case in range(ncases):
        p = int(rng.integers(2, 6))
        ns = [int(rng.integers(p + 1, 12 if rng.random() < 0.8 else 26)) for _ in range(3)]
        mult = rng.random() < 0.2
        kvs = tuple(iga.bspline.make_knots(p, 0.0, 1.0, n, mult=2 if (mult and k == 1 and p > 1) else 1) for k, n in enumerate(ns))
        geo = geos[int(rng.integers(0, len(geos)))]()
        c = [float(x) for x in (1.0 + rng.random(), *(0.3 * rng.standard_normal(3)))]
        if rng.random() < 0.5:
            coeff, cname = iga.assemblers.AffineCoefficient(*c), 'affine'
        else:
            coeff, cname = (lambda x, y, z, c=c: c[0] + c[1] * x * x + c[2] * y + c[3] * z), 'sampled'
        asm = iga.assemblers.ConvDiffAssembler3D(kvs, geo, coeff)
        A = asm.assemble_csr(algo='sumfact')
        path = sorted(asm.patch.last_path())
        E = asm.assemble_csr(algo='entrywise')
        asm.patch.close()
        r = rel(A, E)
        N0 = kvs[0].numdofs
        cut = sorted(set([0, N0] + [int(x) for x in rng.integers(1, max(2, N0), size=2)]))
        parts = []
        for lo, hi in zip(cut[:-1], cut[1:]):
            sl = iga.assemblers.ConvDiffAssembler3D(kvs, geo, coeff, row0=(lo, hi))
            parts.append(sl.assemble_csr(algo='sumfact'))
            sl.patch.close()
        S = scipy.sparse.vstack(parts).tocsr()
        same = np.array_equal(S.data, A.data) and np.array_equal(S.indices, A.indices)
        worst = max(worst, r)
        flag = '' if (r <= 1e-12 and same and not np.isnan(A.data).any()) else '   <-- FAIL'
        print('case %2d p=%d n=%s mult=%d %s %-22s rel %.2e slabs %s%s' % (case, p, ns, mult, cname, '+'.join(path), r, same, flag), flush=True)
    print('worst relative difference %.3e' % worst)


if __name__ == '__main__':
    main()
