#!/usr/bin/env python3
"""Randomised cross-check on the GPU box: forms with second / parametric derivatives (pyiga_amd.pforms -> parametric jet forms in
passes) on the sum-factorised stages against the entry-wise kernel, 2D and 3D, random degrees (2..5), sizes, knot multiplicities,
geometries and random sums of terms.  usage: python3 tools/fuzz_pforms.py [ncases] [seed]"""
import os
import sys

import numpy as np

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import pyiga_amd as iga
from fuzz_paths import random_kv, rel

TERMS2 = ['inner(hess(u), hess(v))', 'div(grad(u)) * div(grad(v))', 'Dx(u, 0, times=2) * Dx(v, 1, times=2)', 'c * tr(hess(u)) * v',
          'inner(b, grad(u)) * v', 'u * Dx(v, 0, parametric=True)', 'hess(u)[0, 1] * v', 'u * hess(v)[1, 1]', 'c * u * v',
          'inner(hess(u, parametric=True), hess(v, parametric=True))', 'inner(grad(u), grad(v))', 'Dx(Dx(u, 0), 1) * Dx(v, 0)']
TERMS3 = TERMS2 + ['Dx(u, 2, times=2) * v', 'hess(u)[0, 2] * Dx(v, 1)', 'Dx(Dx(u, 1), 2) * Dx(Dx(v, 0), 2)', 'u * Dx(v, 2, times=2, parametric=True)']


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    g = iga.geometry
    geos3 = [lambda: g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus()), g.twisted_box]
    geos2 = [g.quarter_annulus, g.bspline_quarter_annulus]
    worst = 0.0
    for case in range(ncases):
        d = 3 if rng.random() < 0.5 else 2
        ps = [int(rng.integers(2, 6)) for _ in range(d)]
        ns = [int(rng.integers(2, 7 if d == 3 else 20)) for _ in range(d)]
        kvs = tuple(random_kv(rng, p, n) for p, n in zip(ps, ns))
        geo = (geos3 if d == 3 else geos2)[int(rng.integers(0, 2))]()
        pool = TERMS3 if d == 3 else TERMS2
        k = int(rng.integers(1, 5))
        terms = [pool[i] for i in rng.choice(len(pool), size=k, replace=False)]
        form = '(' + ' + '.join('%.3f * %s' % (rng.normal(), t) for t in terms) + ') * dx'
        a = rng.normal(size=4)
        inputs = dict(geo=geo, c=(lambda *x: 1.0 + a[0] * x[0] * x[-1]),
                      b=(lambda *x: tuple(a[1 + i] + 0.5 * x[i] + 0.0 * x[0] for i in range(d))))
        asm = iga.assemble.instantiate_assembler(form, kvs, inputs)
        if not isinstance(asm, iga.assemblers._ParametricFormAssembler):
            continue                                       # (first-order terms only: the other front-end)
        S = asm.assemble_csr(algo='sumfact')
        E = asm.assemble_csr(algo='entrywise')
        r = rel(S, E)
        worst = max(worst, r)
        ok = r <= 1e-12 and not np.isnan(S.data).any()
        print('%3d  d=%d p=%s n=%s passes=%d terms=%d  rel %.1e  %s   %s' % (case, d, ps, ns, len(asm.passes), len(asm.terms), r, 'ok' if ok else 'FAIL', form[:90]))
        if not ok:
            sys.exit(1)
    print('all cases ok, worst rel %.2e' % worst)


if __name__ == '__main__':
    main()
