#!/usr/bin/env python3
"""k_bf3 (both triangles from the fused stage) against k_bf2 + mirror pass: same bits expected.  GPU only."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(p, n, kind, bf, geo='cyl'):
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from pyiga_amd import bspline, geometry, assemble
kv = bspline.make_knots(%d, 0.0, 1.0, %d)
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
A = (assemble.stiffness if %r == 'stiffness' else assemble.mass)((kv, kv, kv), geo)
A.sort_indices()
np.save(%r, A.data)
np.save(%r + '.ip.npy', A.indptr)
''' % (ROOT, p, n, kind, '/tmp/bf3_%s.npy' % bf, '/tmp/bf3_%s' % bf)
    env = dict(os.environ)
    if bf == 'old':
        env['IGX_BF'] = '2'
    else:
        env.pop('IGX_BF', None)
    subprocess.check_call([sys.executable, '-c', code], env=env)
    return np.load('/tmp/bf3_%s.npy' % bf), np.load('/tmp/bf3_%s.ip.npy' % bf)


if __name__ == '__main__':
    cases = [(int(a.split(',')[0]), int(a.split(',')[1])) for a in sys.argv[1:]] or [(4, 3), (2, 3), (2, 6), (3, 5), (4, 8), (4, 20), (1, 4)]
    for p, n in cases:
        for kind in ('stiffness', 'mass'):
            a, ip = run(p, n, kind, 'new')
            b, _ = run(p, n, kind, 'old')
            N = n + p
            bad = np.nonzero(a != b)[0]
            print('p=%d n=%d %-9s nnz=%d  mismatching=%d  maxrel=%.3e' % (p, n, kind, a.size, bad.size, abs(a - b).max() / abs(b).max()))
            if bad.size:
                rows = np.searchsorted(ip, bad, side='right') - 1
                for k in bad[:12]:
                    r = np.searchsorted(ip, k, side='right') - 1
                    i0, i1, i2 = r // (N * N), (r // N) % N, r % N
                    print('    row (%d,%d,%d) pos-in-row %d : new %.6e old %.6e' % (i0, i1, i2, k - ip[r], a[k], b[k]))
                ur = np.unique(rows)
                print('    rows affected: %d of %d; i2 values: %s; i1 values: %s; i0 values: %s' % (ur.size, N ** 3, sorted(set((ur % N).tolist()))[:20], sorted(set(((ur // N) % N).tolist()))[:20], sorted(set((ur // (N * N)).tolist()))[:20]))
