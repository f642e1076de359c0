#!/usr/bin/env python3
"""Randomised cross-check on the GPU box: general bilinear forms (IGX_FORM: random sparse tables of coefficient functions in the
jets of u and v) on the sum-factorised stages against the entry-wise kernel, whole patch and row slabs, 2D and 3D, over random
degrees, sizes and knot multiplicities.  usage: python3 tools/fuzz_forms.py [ncases] [seed]"""
import os
import sys

import numpy as np
import scipy.sparse

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import pyiga_amd as iga
from fuzz_paths import random_kv, rel


def coefficient(rng, d):
    a = rng.normal(size=d + 1)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        return float(a[0])
    if kind == 1:
        return lambda *x: a[0] + sum(a[k + 1] * x[k] for k in range(d))
    return lambda *x: 1.0 + a[0] * x[0] * x[-1] + 0.3 * np.sin(a[1] * x[d - 1])


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    g = iga.geometry
    geos3 = [lambda: g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus()), g.twisted_box, lambda: g.unit_cube(3, 2)]
    geos2 = [g.quarter_annulus, g.bspline_quarter_annulus, lambda: g.unit_cube(2, 3)]
    worst = 0.0
    for case in range(ncases):
        d = 3 if rng.random() < 0.4 else 2
        p0 = int(rng.integers(1, 5))
        ps = [p0] * d if rng.random() < 0.5 else [int(rng.integers(1, 5)) for _ in range(d)]
        big = d == 2 and rng.random() < 0.5
        ns = [int(rng.integers(30, 160)) if big else int(rng.integers(2, 12 if d == 3 else 30)) for _ in range(d)]
        kvs = tuple(random_kv(rng, p, n) for p, n in zip(ps, ns))
        geo = (geos3 if d == 3 else geos2)[int(rng.integers(0, 3))]()
        table = [[coefficient(rng, d) if rng.random() < 0.45 else None for _ in range(d + 1)] for _ in range(d + 1)]
        if all(e is None for row in table for e in row):
            table[0][0] = 1.0
        cls = iga.assemblers.GeneralFormAssembler3D if d == 3 else iga.assemblers.GeneralFormAssembler2D
        asm = cls(kvs, geo, table)
        A = asm.assemble_csr(algo='sumfact')
        E = asm.assemble_csr(algo='entrywise')
        r = rel(A, E)
        N0 = kvs[0].numdofs
        cut = sorted(set([0, N0] + [int(x) for x in rng.integers(1, max(2, N0), size=2)]))
        parts = [cls(kvs, geo, table, row0=(lo, hi)).assemble_csr(algo='sumfact') for lo, hi in zip(cut[:-1], cut[1:])]
        S = scipy.sparse.vstack(parts).tocsr()
        slab_ok = np.array_equal(S.indptr, A.indptr) and np.array_equal(S.data, A.data)
        worst = max(worst, r)
        status = 'ok' if (r <= 1e-12 and slab_ok) else 'FAIL'
        print('%3d  d=%d p=%s n=%s terms=%d rel %.1e slabs %s  %s'
              % (case, d, ps, ns, sum(e is not None for row in table for e in row), r, slab_ok, status), flush=True)
        if status != 'ok':
            sys.exit(1)
    print('all %d cases ok, worst rel %.2e' % (ncases, worst))


if __name__ == '__main__':
    main()
