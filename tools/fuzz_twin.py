#!/usr/bin/env python3
"""Randomised cross-check on the GPU box (round 6): 3D patches with repeated knots on the LAST axis only -- assembled through the
axis-exchanged twin (igx_patch::twin; k_bf3 stores to the caller's CSR layout, fused3.hip TR) -- against the entry-wise kernels:
mass, stiffness and the convection-diffusion form, degrees 2..pmax (5) on the mid / last axis, any lower-or-equal degree on axis 0 (with or without repeated knots
there), 2..60 spans on the mid axis (several tiles of the twin's last axis), random multiplicities 1..p on the last axis, five
geometries, row slabs bit for bit, exact symmetry, NaN poison.  usage: python3 tools/fuzz_twin.py [ncases] [seed] [pmax]"""
import os
import sys

import numpy as np
import scipy.sparse

os.environ['IGX_DEBUG_POISON'] = '1'
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import pyiga_amd as iga
from fuzz_paths import random_kv, rel


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    pmax = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    g = iga.geometry
    geos = [lambda: g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus()), g.twisted_box, lambda: g.unit_cube(3, 2),
            lambda: g.tensor_product(g.line_segment(0.0, 2.0, intervals=3), g.bspline_quarter_annulus()),
            lambda: g.tensor_product(g.quarter_annulus(), g.line_segment(0.0, 1.0))]
    worst, ntwin = 0.0, 0
    for case in range(ncases):
        p = int(rng.integers(2, pmax + 1))
        p0 = int(rng.integers(1, p + 1))
        n0, n1, n2 = int(rng.integers(2, 9)), int(rng.integers(2, 61 if rng.random() < 0.4 else 12)), int(rng.integers(2, 12))
        mults = rng.integers(1, p + 1, size=n2 - 1)
        if n2 > 1 and mults.max(initial=1) == 1:
            mults[int(rng.integers(0, n2 - 1))] = 2
        inner = np.repeat(np.arange(1, n2) / n2, mults)
        kv2 = iga.bspline.KnotVector(np.concatenate([np.zeros(p + 1), inner, np.ones(p + 1)]), p)
        kvs = (random_kv(rng, p0, n0), iga.bspline.make_knots(p, 0.0, 1.0, n1), kv2)
        geo = geos[int(rng.integers(0, len(geos)))]()
        kind = ['stiffness', 'mass', 'convdiff'][int(rng.choice(3, p=[0.45, 0.25, 0.3]))]
        if kind == 'convdiff':
            c = [float(x) for x in (1.0 + rng.random(), *(0.3 * rng.standard_normal(3)))]
            coeff = iga.assemblers.AffineCoefficient(*c) if rng.random() < 0.5 else (lambda x, y, z, c=c: c[0] + c[1] * x * x + c[2] * y + c[3] * z)

            def make(**kw):
                asm = iga.assemblers.ConvDiffAssembler3D(kvs, geo, coeff, **kw)
                return asm.patch, (lambda algo: asm.assemble_csr(algo=algo))
        else:
            def make(**kw):
                pt = iga.assemblers.DevicePatch(kvs, geo, **kw)
                return pt, (lambda algo: pt.csr(kind, algo=algo))
        patch, run = make()
        A = run('sumfact')
        path = sorted(patch.last_path())
        E = run('entrywise')
        patch.close()
        r = rel(A, E)
        sym = abs(A - A.T).max() if kind != 'convdiff' else 0.0
        nan = bool(np.isnan(A.data).any())
        N0 = kvs[0].numdofs
        cut = sorted(set([0, N0] + [int(x) for x in rng.integers(1, max(2, N0), size=2)]))
        parts = []
        for lo, hi in zip(cut[:-1], cut[1:]):
            sl, run = make(row0=(lo, hi))
            parts.append(run('sumfact'))
            sl.close()
        S = scipy.sparse.vstack(parts).tocsr()
        slab_ok = np.array_equal(S.indptr, A.indptr) and np.array_equal(S.data, A.data)
        ntwin += 'twin' in path
        worst = max(worst, r)
        want = not (kind == 'convdiff' and p0 < 2)          # (k_geoA has the convection-diffusion form from degree 2 on axis 0)
        status = 'ok' if (r <= 1e-12 and sym == 0.0 and slab_ok and not nan and ('twin' in path) == want) else 'FAIL'
        print('%3d  p=%s N=%s mult(last)=%s %-9s path=%-26s rel %.1e sym %.0e slabs %s  %s'
              % (case, [kv.p for kv in kvs], [kv.numdofs for kv in kvs], list(mults), kind, '+'.join(path) or 'stage', r, sym, slab_ok, status), flush=True)
        if status != 'ok':
            sys.exit(1)
    print('all %d cases ok (%d through the twin), worst rel %.2e' % (ncases, ntwin, worst))


if __name__ == '__main__':
    main()
