import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    """Lazy loader for tests/golden/golden_<name>.npz (made by make_golden.py from the reference)."""
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, 'golden_%s.npz' % name))
        return cache[name]
    return load


@pytest.fixture(scope='session')
def oracle():
    from oracle import iga_oracle
    iga_oracle.build()
    return iga_oracle


def golden_csr(g, name):
    import scipy.sparse
    shape = tuple(int(x) for x in g[name + '_shape'])
    return scipy.sparse.csr_matrix((g[name + '_data'], g[name + '_indices'], g[name + '_indptr']), shape=shape)


def rel_maxdiff(A, B):
    """max_ij |A-B| / max_ij |B|  -- the parity norm of SURVEY section 8c."""
    D = abs(A - B)
    return (D.max() if D.nnz else 0.0) / abs(B).max()


# ---------------------------------------------------------------------------------------------
# general form strings: the inputs and, written out by hand, the coefficient tables
# P[r][s] of  sum_rs P_rs D_r v D_s u  (D_0 = id, D_1..3 = d/dx, d/dy, d/dz) the strings stand for
def form_inputs():
    import numpy as np

    def c(x, y, z):
        return 1.0 + x * y

    def K(x, y, z):
        one = np.ones_like(x * y * z)
        rows = (((2.0 + x) * one, 0.3 * y * one, 0.0 * one), (-0.2 * one, (1.0 + z) * one, 0.1 * x * one),
                (0.5 * one, 0.0 * one, 3.0 * one))
        return np.stack([np.stack(r, -1) for r in rows], -2)

    def b(x, y, z):
        one = np.ones_like(x * y * z)
        return (x * one, (1 + z) * one, y * one)
    return dict(c=c, K=K, b=b)


FORMS = {
    'reactdiff': ('(inner(grad(u), grad(v)) + c*u*v) * dx', ('c',)),
    'aniso': ('inner(dot(K, grad(u)), grad(v)) * dx', ('K',)),
    'adjconv': ('(u * inner(b, grad(v)) + 2.5 * u * v) * dx', ('b',)),
    'full': ('(inner(dot(K, grad(u)), grad(v)) + inner(b, grad(u)) * v + u * inner(b, grad(v)) + c * u * v) * dx', ('K', 'b', 'c')),
    'scaled': ('(0.5 * inner(grad(u), 3 * grad(v)) - inner((x[2], 0.0, -x[0]), grad(u)) * v / 4) * dx', ()),
}


def form_tables():
    inp = form_inputs()
    Kf, bf, cf = inp['K'], inp['b'], inp['c']

    def Kc(i, j):
        return lambda x, y, z: Kf(x, y, z)[..., i, j]

    def bc(i):
        return lambda x, y, z: bf(x, y, z)[i]

    def empty():
        return [[None] * 4 for _ in range(4)]
    T = {}
    t = empty()
    for k in range(1, 4):
        t[k][k] = 1.0
    t[0][0] = cf
    T['reactdiff'] = t
    t = empty()
    for i in range(3):
        for j in range(3):
            t[1 + i][1 + j] = Kc(i, j)
    T['aniso'] = t
    t = empty()
    for i in range(3):
        t[1 + i][0] = bc(i)
    t[0][0] = 2.5
    T['adjconv'] = t
    t = empty()
    for i in range(3):
        for j in range(3):
            t[1 + i][1 + j] = Kc(i, j)
        t[0][1 + i] = bc(i)
        t[1 + i][0] = bc(i)
    t[0][0] = cf
    T['full'] = t
    t = empty()
    for k in range(1, 4):
        t[k][k] = 1.5
    t[0][1] = lambda x, y, z: -z / 4
    t[0][3] = lambda x, y, z: x / 4
    T['scaled'] = t
    return T


def form2d_cases():
    """2D forms of the golden file: name -> (string, inputs, hand-written 3x3 coefficient table)."""
    import numpy as np

    def K2(x, y):
        one = np.ones_like(x * y)
        return np.stack([np.stack(((1.5 + y) * one, 0.4 * x * one), -1), np.stack((-0.3 * one, (2.0 + x * y) * one), -1)], -2)

    def b2(x, y):
        one = np.ones_like(x * y)
        return (y * one, (1.0 - x) * one)

    def cc(x, y):
        return 1.0 + x * y
    t1 = [[cc, None, None], [None, 1.0, None], [None, None, 1.0]]
    t2 = [[3.0, lambda x, y: b2(x, y)[0], lambda x, y: b2(x, y)[1]],
          [lambda x, y: -b2(x, y)[0], lambda x, y: K2(x, y)[..., 0, 0], lambda x, y: K2(x, y)[..., 0, 1]],
          [lambda x, y: -b2(x, y)[1], lambda x, y: K2(x, y)[..., 1, 0], lambda x, y: K2(x, y)[..., 1, 1]]]
    return {'reactdiff': ('(inner(grad(u), grad(v)) + c*u*v) * dx', dict(c=cc), t1),
            'full': ('(inner(dot(K, grad(u)), grad(v)) + inner(b, grad(u)) * v - u * inner(b, grad(v)) + 3 * u * v) * dx', dict(K=K2, b=b2), t2)}


# forms with second derivatives / parametric derivatives (golden_pforms.npz: the strings make_golden.py gave the reference)
PFORMS2 = {
    'biharm': ('inner(hess(u), hess(v)) * dx', ()),
    'laplap': ('div(grad(u)) * div(grad(v)) * dx', ()),
    'dxx_dyy': ('Dx(u, 0, times=2) * Dx(v, 1, times=2) * dx', ()),
    'pgrad': ('inner(grad(u, parametric=True), grad(v, parametric=True)) * dx', ()),
    'phess': ('inner(hess(u, parametric=True), hess(v, parametric=True)) * dx', ()),
    'mixed': ('(c * tr(hess(u)) * v + inner(b, grad(u)) * v + 0.5 * inner(hess(u), hess(v)) + u * Dx(v, 0, parametric=True)) * dx', ('c', 'b')),
    'hxy': ('(hess(u)[0, 1] * v - u * hess(v)[1, 1]) * dx', ()),
    'khess': ('inner(dot(K, grad(u)), grad(v)) * dx + tr(dot(K, hess(u))) * v * dx', ('K',)),
}
PFORMS3 = {
    'biharm': ('inner(hess(u), hess(v)) * dx', ()),
    'laplap': ('div(grad(u)) * div(grad(v)) * dx', ()),
    'dxx_dzz': ('Dx(u, 0, times=2) * Dx(v, 2, times=2) * dx', ()),
    'phess': ('inner(hess(u, parametric=True), grad(grad(v, parametric=True), parametric=True)) * dx', ()),
    'mixed': ('(c * tr(hess(u)) * v + inner(b, grad(u)) * v + 0.5 * inner(hess(u), hess(v)) + u * Dx(v, 1, parametric=True)) * dx', ('c', 'b')),
    'hxz': ('(hess(u)[0, 2] * v - Dx(Dx(u, 1), 2) * Dx(v, 0)) * dx', ()),
}


def pform_inputs2():
    import numpy as np

    def K2(x, y):
        one = np.ones_like(x * y)
        return np.stack([np.stack(((1.5 + y) * one, 0.4 * x * one), -1), np.stack((-0.3 * one, (2.0 + x * y) * one), -1)], -2)

    def b2(x, y):
        one = np.ones_like(x * y)
        return (y * one, (1.0 - x) * one)
    return dict(c=lambda x, y: 1.0 + x * y, b=b2, K=K2)
