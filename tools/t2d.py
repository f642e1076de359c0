import os, numpy as np, sys
sys.path.insert(0, '.')
import pyiga_amd as iga
def run(p, n, path, env=None):
    os.environ['IGX_PATH'] = path
    for k in ('IGX_BF', 'IGX_BF_MCHUNKS'): os.environ.pop(k, None)
    if env: os.environ.update(env)
    kv = iga.bspline.make_knots(p, 0., 1., n)
    return iga.assemble.stiffness((kv, kv), iga.geometry.quarter_annulus())
for p, n, env in ((3, 40, None), (3, 40, {'IGX_BF_MCHUNKS': '1'}), (3, 20, None), (3, 20, {'IGX_BF_MCHUNKS': '1'}), (3, 27, {'IGX_BF_MCHUNKS': '1'})):
    A0 = run(p, n, 'unfused'); A2 = run(p, n, 'fused', env)
    D = abs(A2 - A0).tocoo()
    bad = ~(D.data <= 1e-12 * abs(A0).max())
    print(p, n, env, 'bad entries', bad.sum(), 'of', A0.nnz)
    if bad.sum():
        r, c = D.row[bad], D.col[bad]
        N1 = n + p
        print('   rows i0 (mid):', sorted(set((r // N1).tolist())))
        print('   rows i1 (last):', sorted(set((r % N1).tolist())))
        low = (c <= r)
        rl, cl = r[low], c[low]
        print('   lower: j1-i1:', sorted(set(((cl % N1) - (rl % N1)).tolist())), ' j0-i0:', sorted(set(((cl // N1) - (rl // N1)).tolist())))
        i0 = 10
        sel = low & (r // N1 == i0)
        print('   row i0=10 lower bad (i1, j0-i0, j1-i1):', sorted(set(zip((r[sel] % N1).tolist(), ((c[sel] // N1) - i0).tolist(), ((c[sel] % N1) - (r[sel] % N1)).tolist())))[:60])
