#!/usr/bin/env python3
"""Randomised cross-check on the GPU box of the load vector (k_lv12 + axis-0 contraction, or the three separate contractions
where the library chooses them): device against a numpy restatement of the tensor contraction on the same Gauss grid
(W from the device's mass field), repeatable bit for bit, slab by slab.  usage: python3 tools/fuzz_rhs.py [ncases] [seed]"""
import sys

import numpy as np

sys.path.insert(0, '.')
import pyiga_amd as iga


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    g = iga.geometry
    geos = [lambda: g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus()), g.twisted_box, lambda: g.unit_cube()]
    worst = 0.0
    for case in range(ncases):
        same = rng.random() < 0.7
        p0 = int(rng.integers(1, 6))
        ps = [p0] * 3 if same else [int(rng.integers(1, 6)) for _ in range(3)]
        ns = [int(rng.integers(2, 9)), int(rng.integers(3, 70)), int(rng.integers(2, 30))]
        kvs = tuple(iga.bspline.make_knots(p, 0.0, 1.0, n) for p, n in zip(ps, ns))
        geo = geos[int(rng.integers(0, 3))]()
        f = lambda x, y, z: np.cos(3 * x) * np.exp(y) + z * z
        patch = iga.assemblers.DevicePatch(kvs, geo)
        grid = tuple(patch.gauss(a)[0] for a in range(3))
        fvals = iga.utils.grid_eval_transformed(f, grid, geo)
        b = patch.load_vector(fvals)
        b2 = patch.load_vector(fvals)
        W = patch.fields('mass')[0]
        patch.close()
        # numpy: collocation matrices of the active functions, contraction axis by axis
        t = fvals * W
        for k, kv in enumerate(kvs):
            C = iga.bspline.collocation(kv, grid[k]).toarray() if hasattr(iga.bspline, 'collocation') else None
            if C is None:
                break
            t = np.moveaxis(np.tensordot(C.T, np.moveaxis(t, k, 0), axes=1), 0, k)
        r = float(np.abs(b - t).max() / np.abs(t).max()) if C is not None else float('nan')
        N0 = kvs[0].numdofs
        cut = sorted(set([0, N0, int(rng.integers(1, max(2, N0)))]))
        parts = []
        for lo, hi in zip(cut[:-1], cut[1:]):
            sl = iga.assemblers.DevicePatch(kvs, geo, row0=(lo, hi))
            parts.append(sl.load_vector(fvals))
            sl.close()
        same_slabs = np.array_equal(np.concatenate(parts, axis=0), b)
        ok = np.array_equal(b, b2) and same_slabs and (np.isnan(r) or r <= 1e-12)
        worst = max(worst, 0.0 if np.isnan(r) else r)
        print('case %2d p=%s n=%s rel %.2e repeat %s slabs %s%s' % (case, ps, ns, r, np.array_equal(b, b2), same_slabs, '' if ok else '   <-- FAIL'), flush=True)
    print('worst relative difference %.3e' % worst)


if __name__ == '__main__':
    main()
