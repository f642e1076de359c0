#!/usr/bin/env python3
"""Randomised cross-check on the GPU box: default sum-factorised chain (k_geoA / k_bf / k_mirror or the stage kernels, as the
library chooses) against the entry-wise kernels, full patch and row slabs, over random degrees, sizes, knot multiplicities
and geometries.  usage: python3 tools/fuzz_paths.py [ncases] [seed] [big2d|-] [pmax]   (big2d: 2D patches of 40-300 spans per axis only --
the stage kernels of 2D instead of the single launch)"""
import sys

import numpy as np
import scipy.sparse

sys.path.insert(0, '.')
import pyiga_amd as iga


def rel(A, B):
    D = abs(A - B)
    return (D.max() if D.nnz else 0.0) / abs(B).max()


def random_kv(rng, p, n):
    kv = iga.bspline.make_knots(p, 0.0, 1.0, n)
    if p >= 2 and rng.random() < 0.3 and n >= 4:        # repeat a few interior knots
        knots = list(kv.kv)
        for u in rng.choice(np.unique(kv.kv)[1:-1], size=min(2, n - 1), replace=False):
            knots.extend([u] * int(rng.integers(1, p)))
        kv = iga.bspline.KnotVector(np.sort(np.array(knots)), p)
    return kv


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    g = iga.geometry
    geos3 = [lambda: g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus()), g.twisted_box, lambda: g.unit_cube(3, 2),
             lambda: g.tensor_product(g.line_segment(0.0, 2.0, intervals=3), g.bspline_quarter_annulus()),
             lambda: g.tensor_product(g.quarter_annulus(), g.line_segment(0.0, 1.0))]     # degree 2 (NURBS) along axis 0
    geos2 = [g.quarter_annulus, g.bspline_quarter_annulus, lambda: g.unit_cube(2, 3)]
    big2d = len(sys.argv) > 3 and sys.argv[3] == 'big2d'
    pmax = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    worst = 0.0
    for case in range(ncases):
        d = 2 if big2d else 3 if rng.random() < 0.7 else 2
        same = rng.random() < 0.5
        p0 = int(rng.integers(1, pmax + 1))
        ps = [p0] * d if same else [int(rng.integers(1, pmax + 1)) for _ in range(d)]
        ns = [int(rng.integers(40, 300)) if big2d else int(rng.integers(2, 14 if d == 3 else 40)) for _ in range(d)]
        kvs = tuple(random_kv(rng, p, n) for p, n in zip(ps, ns))
        geo = (geos3 if d == 3 else geos2)[int(rng.integers(0, len(geos3) if d == 3 else 3))]()
        kind = 'stiffness' if rng.random() < 0.7 else 'mass'
        patch = iga.assemblers.DevicePatch(kvs, geo)
        A = patch.csr(kind, algo='sumfact')
        path = sorted(patch.last_path())
        E = patch.csr(kind, algo='entrywise')
        patch.close()
        r = rel(A, E)
        sym = abs(A - A.T).max()
        N0 = kvs[0].numdofs
        cut = sorted(set([0, N0] + [int(x) for x in rng.integers(1, max(2, N0), size=2)]))
        parts = []
        for lo, hi in zip(cut[:-1], cut[1:]):
            sl = iga.assemblers.DevicePatch(kvs, geo, row0=(lo, hi))
            parts.append(sl.csr(kind, algo='sumfact'))
            sl.close()
        S = scipy.sparse.vstack(parts).tocsr()
        slab_ok = np.array_equal(S.indptr, A.indptr) and np.array_equal(S.data, A.data)
        worst = max(worst, r)
        status = 'ok' if (r <= 1e-12 and sym == 0.0 and slab_ok) else 'FAIL'
        print('%3d  d=%d p=%s n=%s mult=%s %-9s path=%-28s rel %.1e sym %.0e slabs %s  %s'
              % (case, d, ps, ns, [int(kv.kv.size - np.unique(kv.kv).size - 2 * kv.p) for kv in kvs], kind, '+'.join(path) or 'stage', r, sym, slab_ok, status), flush=True)
        if status != 'ok':
            sys.exit(1)
    print('all %d cases ok, worst rel %.2e' % (ncases, worst))


if __name__ == '__main__':
    main()
