#!/usr/bin/env python3
"""Randomised cross-check on the GPU box (round 6): first-order form STRINGS -- random combinations of a diffusion part (a scalar
function, a constant symmetric or non-symmetric tensor), convection terms both ways and a reaction term, coefficients constants or
traced functions -- assembled on the fast chain (k_geoA<FORM = 2 | 3> evaluates the coefficient table inside the axis-0 sweep, k_bf3
finishes) against the entry-wise kernel; whole patch and row slabs (bit for bit), exact symmetry where the table is symmetric, which
chain ran.  Random degrees 2..5 (equal on the mid / last axis for non-symmetric tables), sizes, knot multiplicities on axes 0 and 1,
four geometries (NURBS / B-spline, degree 1 and 2 along axis 0).  usage: python3 tools/fuzz_form_tables.py [ncases] [seed]"""
import os
import sys

import numpy as np
import scipy.sparse

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import pyiga_amd as iga
from fuzz_paths import random_kv, rel


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    g = iga.geometry
    geos = [lambda: g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus()), g.twisted_box,
            lambda: g.tensor_product(g.quarter_annulus(), g.line_segment(0.0, 1.0)),
            lambda: g.tensor_product(g.bspline_quarter_annulus(), g.line_segment(0.0, 1.5, intervals=2))]
    worst, on_chain = 0.0, 0
    for case in range(ncases):
        a = rng.normal(size=8)
        inputs = {'f1': lambda x, y, z: 1.5 + a[0] * x + 0.3 * np.sin(a[1] * z), 'f2': lambda x, y, z: a[2] + a[3] * x * y,
                  'f3': lambda x, y, z: 2.0 + np.cos(a[4] * y)}

        def coef():
            return str(round(float(rng.normal()), 3)) if rng.random() < 0.5 else ['f1', 'f2', 'f3'][int(rng.integers(0, 3))]

        def vec():
            return '(%s, %s, %s)' % (coef(), coef(), coef())
        parts, sym = [], True
        kd = int(rng.integers(0, 4))
        if kd == 0:
            parts.append('%s * inner(grad(u), grad(v))' % coef())
        elif kd == 1:
            m = rng.normal(size=(3, 3)); m = m + m.T + 4 * np.eye(3)
            parts.append('inner(dot(%s, grad(u)), grad(v))' % str(tuple(tuple(round(float(v), 3) for v in row) for row in m)))
        elif kd == 2:
            m = rng.normal(size=(3, 3)) + 3 * np.eye(3)
            parts.append('inner(dot(%s, grad(u)), grad(v))' % str(tuple(tuple(round(float(v), 3) for v in row) for row in m)))
            sym = False
        if rng.random() < 0.5:
            parts.append('inner(%s, grad(u)) * v' % vec()); sym = False
        if rng.random() < 0.25:
            parts.append('u * inner(%s, grad(v))' % vec()); sym = False
        if rng.random() < 0.6 or not parts:
            parts.append('%s * u * v' % coef())
        form = '(' + ' + '.join(parts) + ') * dx'
        p0 = int(rng.integers(2, 6))
        p12 = int(rng.integers(2, 6))
        ps = [p0, p12, p12]
        if sym and rng.random() < 0.4:
            ps[1 + int(rng.integers(0, 2))] = max(2, max(ps) - 1) if max(ps) > 2 else ps[1]
        ns = [int(rng.integers(2, 9)) for _ in range(3)]
        kvs = (random_kv(rng, ps[0], ns[0]), random_kv(rng, ps[1], ns[1]), iga.bspline.make_knots(ps[2], 0.0, 1.0, ns[2]))
        if ps[1] == ps[2] and rng.random() < 0.3:        # repeated knots on the LAST axis only: through the twin patch
            kvs = (kvs[0], iga.bspline.make_knots(ps[1], 0.0, 1.0, ns[1]), iga.bspline.make_knots(ps[2], 0.0, 1.0, max(2, ns[2]), mult=int(rng.integers(2, ps[2] + 1))))
        geo = geos[int(rng.integers(0, 4))]()
        cls = iga.assemblers.GeneralFormAssembler3D
        asm = cls(kvs, geo, form, inputs=inputs)
        A = asm.assemble_csr(algo='sumfact')
        path = asm.patch.last_path()
        E = asm.assemble_csr(algo='entrywise')
        r = rel(A, E)
        N0 = kvs[0].numdofs
        cut = sorted(set([0, N0] + [int(x) for x in rng.integers(1, max(2, N0), size=2)]))
        parts_ = [cls(kvs, geo, form, inputs=inputs, row0=(lo, hi)).assemble_csr(algo='sumfact') for lo, hi in zip(cut[:-1], cut[1:])]
        S = scipy.sparse.vstack(parts_).tocsr()
        slab_ok = np.array_equal(S.indptr, A.indptr) and np.array_equal(S.data, A.data)
        fast = 'geoA' in path and 'bf3' in path
        on_chain += fast
        sym_ok = (abs(A - A.T).max() == 0.0) if (fast and 'both' in path) else True
        worst = max(worst, r)
        status = 'ok' if (r <= 1e-12 and slab_ok and sym_ok) else 'FAIL'
        if kvs[2].numdofs > kvs[2].numspans + kvs[2].p and 'twin' not in path:
            # repeated knots on the last axis and not through the twin: then the form must be one the fast chain does not take with
            # the same knots on the MID axis either (a non-symmetric diffusion tensor, degrees)
            sw = cls((kvs[0], kvs[2], kvs[1]), geo, form, inputs=inputs)
            sw.assemble_csr(algo='sumfact')
            if 'bf3' in sw.patch.last_path() and 'geoA' in sw.patch.last_path():
                status = 'FAIL (no twin)'
            sw.patch.close()
        print('%3d  p=%s N=%s %-9s rel %.1e slabs %s sym %s  %s  %s' % (case, ps, [kv.numdofs for kv in kvs], '+'.join(sorted(path)) or 'stages', r, slab_ok, sym_ok, status, form[:90]), flush=True)
        if status != 'ok':
            sys.exit(1)
    print('all %d cases ok (%d on the fast chain), worst rel %.2e' % (ncases, on_chain, worst))


if __name__ == '__main__':
    main()
