"""Form strings with vector-valued basis functions and boundary integrals (SURVEY section 8, row f1 "full").

The reference turns such a string into a ``VForm`` (pyiga/vform.py:1804-1885 ``parse_vf`` with ``bfuns=[('u', 2), ...]``,
``boundary=True``), compiles a kernel per form (pyiga/codegen/cython.py) and drives it block-wise
(pyiga/assemble.py:760-811 ``assemble_entries_vec``, pyiga/genericasm.pxi ``multi_blocks``).  Here the string is evaluated
once, numerically, on the Gauss grid with the tensor-valued expressions below; the outcome is, for every pair of
components (a of v, b of u), the table of physical coefficient fields

    a(u, v) = integral of  sum_{a,b} sum_{r,s=0..d}  P[a][b][r][s](x) * D_r v_a * D_s u_b ,     D_0 = id, D_1..d = d/dx, d/dy[, d/dz]

i.e. one scalar jet form (``IGX_FORM`` of libigx, include/igx.h) per block, which the device assembles.  An expression is
an array over the grid whose trailing axes are  tensor shape  +  (component, jet) of the basis function(s) it is linear in:

    coefficient                        G + shape
    linear in v or u                   G + shape + (nc, d + 1)
    bilinear (v first, then u)         G + ()    + (ncv, d + 1, ncu, d + 1)

Names: ``u, v`` (or the names given in `bfuns`), ``grad div curl inner dot cross outer tr``, ``as_vector as_matrix``,
``.dot() .T [k]``, ``+ - * /``, ``x`` (physical coordinates), ``n`` (outer unit normal, boundary integrals), ``dx``, ``ds``,
numbers, tuples, named inputs (callables of the physical coordinates, constants, arrays).
"""
import numpy as np


class TExpr:
    __array_priority__ = 1000

    def __init__(self, a, shape, kind, who=None, const=False):
        self.a, self.shape, self.kind, self.who, self.const = a, tuple(shape), kind, who, const
        # kind: 'c' coefficient, 'l' linear in basis function `who` = (name, is_test_function), 'b' bilinear (test function first)

    # ---- helpers
    @property
    def tail(self):
        return {'c': 0, 'l': 2, 'b': 4}[self.kind]

    def _grid_ndim(self):
        return self.a.ndim - len(self.shape) - self.tail

    def _expand(self, shape_nd, tail_front, tail_back):
        """View with the tensor axes padded on the left to `shape_nd` axes and `tail_front` / `tail_back` unit axes inserted
        before / after the own tail (to line a 'l' operand up inside a 'b' result)."""
        g = self._grid_ndim()
        a = self.a
        sh = a.shape
        new = sh[:g] + (1,) * (shape_nd - len(self.shape)) + sh[g:g + len(self.shape)] + (1,) * tail_front + sh[g + len(self.shape):] + (1,) * tail_back
        return a.reshape(new)

    @staticmethod
    def wrap(x):
        if isinstance(x, TExpr):
            return x
        if isinstance(x, (tuple, list)):
            return stack([TExpr.wrap(c) for c in x])
        a = np.asarray(x, dtype=float)
        return TExpr(a, a.shape, 'c', const=True)

    # ---- algebra
    def _mul(self, o):
        o = TExpr.wrap(o)
        if self.shape and o.shape and self.shape != o.shape:
            raise NotImplementedError('elementwise product of tensors of different shape %s, %s: use dot() or inner()' % (self.shape, o.shape))
        nd = max(len(self.shape), len(o.shape))
        kinds = self.kind + o.kind
        const = self.const and o.const
        shape = self.shape or o.shape
        if kinds == 'cc':
            return TExpr(self._expand(nd, 0, 0) * o._expand(nd, 0, 0), shape, 'c', const=const)
        if kinds in ('cl', 'lc'):
            c, l = (self, o) if self.kind == 'c' else (o, self)
            return TExpr(c._expand(nd, 0, 2) * l._expand(nd, 0, 0), shape, 'l', l.who, const)
        if kinds in ('cb', 'bc'):
            c, b = (self, o) if self.kind == 'c' else (o, self)
            if nd:
                raise NotImplementedError('a bilinear integrand must be scalar')
            return TExpr(c._expand(0, 0, 4) * b.a, (), 'b')
        if kinds == 'll':
            if self.who == o.who:
                raise NotImplementedError('the form is not bilinear: two factors contain %s' % self.who[0])
            t, s = (self, o) if self.who[1] else (o, self)       # test function first
            return TExpr(t._expand(nd, 0, 2) * s._expand(nd, 2, 0), shape, 'b')
        raise NotImplementedError('the form is not bilinear')

    def __mul__(self, o):
        if isinstance(o, _Measure):
            return o.__rmul__(self)
        return self._mul(o)
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = TExpr.wrap(o)
        if o.kind != 'c' or o.shape:
            raise NotImplementedError('division by anything but a scalar coefficient')
        return self._mul(TExpr(1.0 / o.a, (), 'c', const=o.const))

    def __rtruediv__(self, o):
        if self.kind != 'c':
            raise NotImplementedError('division by a basis-function expression')
        return TExpr.wrap(o)._mul(TExpr(1.0 / self.a, self.shape, 'c', const=self.const))

    def __neg__(self):
        return TExpr(-self.a, self.shape, self.kind, self.who, self.const)

    def __pos__(self):
        return self

    def __pow__(self, k):
        if self.kind != 'c':
            raise NotImplementedError('power of a basis-function expression')
        return TExpr(self.a ** k, self.shape, 'c', const=self.const)

    def __add__(self, o):
        o = TExpr.wrap(o)
        if self.kind != o.kind or self.who != o.who or (self.shape != o.shape and not (self.kind == 'c' and (not self.shape or not o.shape))):
            raise NotImplementedError('sum of incompatible expressions')
        nd = max(len(self.shape), len(o.shape))
        return TExpr(self._expand(nd, 0, 0) + o._expand(nd, 0, 0), self.shape or o.shape, self.kind, self.who, self.const and o.const)
    __radd__ = __add__

    def __sub__(self, o):
        return self + (-TExpr.wrap(o))

    def __rsub__(self, o):
        return TExpr.wrap(o) + (-self)

    def __getitem__(self, k):
        k = k if isinstance(k, tuple) else (k,)
        if len(k) > len(self.shape) or any(not isinstance(i, (int, np.integer)) for i in k):
            raise NotImplementedError('only integer component indices are supported')
        g = self._grid_ndim()
        return TExpr(self.a[(slice(None),) * g + tuple(int(i) for i in k)], self.shape[len(k):], self.kind, self.who, self.const)

    def __iter__(self):
        if not self.shape:
            raise TypeError('scalar expression is not iterable')
        return iter(self[k] for k in range(self.shape[0]))

    def __len__(self):
        return self.shape[0]

    @property
    def T(self):
        if len(self.shape) != 2:
            raise NotImplementedError('.T of a non-matrix')
        g = self._grid_ndim()
        return TExpr(np.swapaxes(self.a, g, g + 1), self.shape[::-1], self.kind, self.who, self.const)

    def dot(self, o):
        return dot(self, o)


def stack(items):
    """Vector / matrix literal from component expressions."""
    items = [TExpr.wrap(i) for i in items]
    kinds = {i.kind for i in items} - {'c'}
    if len(kinds) > 1 or len({i.who for i in items if i.kind != 'c'}) > 1 or len({i.shape for i in items}) != 1:
        raise NotImplementedError('tuple of incompatible expressions')
    kind = kinds.pop() if kinds else 'c'
    if kind == 'b':
        raise NotImplementedError('tuple of bilinear expressions')
    ref = next((i for i in items if i.kind == kind))
    lifted = []
    for i in items:
        if i.kind != kind:                # a coefficient among linear expressions: only 0 is meaningful
            if np.any(i.a != 0.0):
                raise NotImplementedError('tuple mixing coefficients and basis-function expressions')
            i = TExpr(np.zeros((1,) * ref.a.ndim), ref.shape, kind, ref.who, True)
        lifted.append(i)
    g = max(i._grid_ndim() for i in lifted)
    arrs = []
    for i in lifted:
        a = i.a.reshape((1,) * (g - i._grid_ndim()) + i.a.shape)
        arrs.append(a)
    full = np.broadcast_shapes(*(a.shape for a in arrs))
    out = np.stack([np.broadcast_to(a, full) for a in arrs], axis=g)
    return TExpr(out, (len(items),) + ref.shape, kind, ref.who, all(i.const for i in lifted))


def _sum_axes(e, naxes):
    """Sum over the first `naxes` tensor axes."""
    g = e._grid_ndim()
    return TExpr(e.a.sum(axis=tuple(range(g, g + naxes))), e.shape[naxes:], e.kind, e.who, e.const)


def inner(a, b):
    a, b = TExpr.wrap(a), TExpr.wrap(b)
    if a.shape != b.shape:
        raise NotImplementedError('inner() of tensors of different shape %s, %s' % (a.shape, b.shape))
    return _sum_axes(a * b, len(a.shape)) if a.shape else a * b


def dot(a, b):
    """Contraction of the last axis of `a` with the first axis of `b`."""
    a, b = TExpr.wrap(a), TExpr.wrap(b)
    if not a.shape or not b.shape:
        return a * b
    if a.shape[-1] != b.shape[0]:
        raise NotImplementedError('dot() of shapes %s, %s' % (a.shape, b.shape))
    m = a.shape[-1]
    ga = a._grid_ndim()
    total = None
    for k in range(m):
        ak = TExpr(a.a[(slice(None),) * (ga + len(a.shape) - 1) + (k,)], a.shape[:-1], a.kind, a.who, a.const)
        bk = b[k]
        # outer product of the remaining axes
        term = _outer(ak, bk)
        total = term if total is None else total + term
    return total


def _outer(a, b):
    if not a.shape or not b.shape:
        return a * b
    ga, gb = a._grid_ndim(), b._grid_ndim()
    aa = TExpr(a.a.reshape(a.a.shape[:ga + len(a.shape)] + (1,) * len(b.shape) + a.a.shape[ga + len(a.shape):]), a.shape + (1,) * len(b.shape), a.kind, a.who, a.const)
    bb = TExpr(b.a.reshape(b.a.shape[:gb] + (1,) * len(a.shape) + b.a.shape[gb:]), (1,) * len(a.shape) + b.shape, b.kind, b.who, b.const)
    # elementwise product with broadcasting over the unit axes
    r = TExpr(aa.a, (), aa.kind, aa.who, aa.const)._mul_raw(TExpr(bb.a, (), bb.kind, bb.who, bb.const), len(a.shape) + len(b.shape))
    r.shape = a.shape + b.shape
    return r


def _mul_raw(self, o, nd):
    """Product of two expressions whose arrays already carry `nd` aligned tensor axes (unit axes broadcast)."""
    g1, g2 = self.a.ndim - nd - self.tail, o.a.ndim - nd - o.tail
    g = max(g1, g2)
    A = self.a.reshape((1,) * (g - g1) + self.a.shape)
    B = o.a.reshape((1,) * (g - g2) + o.a.shape)
    kinds = self.kind + o.kind
    if kinds == 'cc':
        return TExpr(A * B, (), 'c', const=self.const and o.const)
    if kinds == 'cl':
        return TExpr(A[..., None, None] * B, (), 'l', o.who, self.const and o.const)
    if kinds == 'lc':
        return TExpr(A * B[..., None, None], (), 'l', self.who, self.const and o.const)
    if kinds == 'll':
        if self.who == o.who:
            raise NotImplementedError('the form is not bilinear: two factors contain %s' % self.who[0])
        if self.who[1]:
            return TExpr(A[..., None, None] * B[..., None, None, :, :], (), 'b')
        return TExpr(B[..., None, None] * A[..., None, None, :, :], (), 'b')
    raise NotImplementedError('the form is not bilinear')


TExpr._mul_raw = _mul_raw


def outer(a, b):
    return _outer(TExpr.wrap(a), TExpr.wrap(b))


def cross(a, b):
    a, b = TExpr.wrap(a), TExpr.wrap(b)
    if a.shape == (3,) and b.shape == (3,):
        return stack([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])
    if a.shape == (3,) and b.shape[:1] == (3,) and len(b.shape) == 2:          # column-wise
        return stack([cross(a, b.T[k]) for k in range(b.shape[1])]).T
    if a.shape == (2,) and b.shape == (2,):
        return a[0] * b[1] - a[1] * b[0]
    raise NotImplementedError('cross() of shapes %s, %s' % (a.shape, b.shape))


def tr(m):
    m = TExpr.wrap(m)
    if len(m.shape) != 2 or m.shape[0] != m.shape[1]:
        raise NotImplementedError('tr() of a non-square tensor')
    total = None
    for k in range(m.shape[0]):
        total = m[k, k] if total is None else total + m[k, k]
    return total


def make_grad(d):
    def grad(e):
        e = TExpr.wrap(e)
        if e.kind != 'l' or not e.const:
            raise NotImplementedError('grad() of anything but a basis function (times constants)')
        g = e._grid_ndim()
        a = e.a
        if np.any(a[..., 1:] != 0.0):
            raise NotImplementedError('second derivatives of basis functions')
        out = np.zeros(a.shape[:g + len(e.shape)] + (d,) + a.shape[g + len(e.shape):])
        for k in range(d):
            out[(slice(None),) * (g + len(e.shape)) + (k, slice(None), 1 + k)] = a[..., 0]
        return TExpr(out, e.shape + (d,), 'l', e.who, True)
    return grad


class _Measure:
    def __init__(self, name):
        self.name = name

    def __rmul__(self, o):
        o = TExpr.wrap(o) if not isinstance(o, Measured) else o
        if isinstance(o, Measured):
            raise NotImplementedError('measure applied twice')
        if o.shape or o.kind == 'c':
            raise NotImplementedError('%s must multiply a scalar integrand in the basis functions' % self.name)
        return Measured(o, self.name)
    __mul__ = __rmul__


class Measured:
    def __init__(self, e, measure):
        self.e, self.measure = e, measure

    def _same(self, o):
        if not isinstance(o, Measured) or o.measure != self.measure or o.e.kind != self.e.kind:
            raise NotImplementedError('sum of incompatible integrals (is every term multiplied by the same measure?)')

    def __add__(self, o):
        self._same(o)
        return Measured(self.e + o.e, self.measure)

    def __sub__(self, o):
        self._same(o)
        return Measured(self.e - o.e, self.measure)

    def __neg__(self):
        return Measured(-self.e, self.measure)

    def __mul__(self, c):
        return Measured(self.e * c, self.measure)
    __rmul__ = __mul__

    def __truediv__(self, c):
        return Measured(self.e / c, self.measure)


def normalise_bfuns(expr, bfuns):
    """[(name, components)] in the order (trial, test) the reference uses: names sorted ('u' before 'v') when detected from
    the string (pyiga/vform.py:1822-1825), else as given."""
    import re
    if bfuns is None:
        words = set(re.findall(r"[^\d\W]\w*", expr))
        return [(bf, 1) for bf in sorted(words & {'u', 'v'})]
    out = []
    for bf in bfuns:
        if isinstance(bf, str):
            bf = (bf,)
        bf = tuple(bf)
        if len(bf) == 1:
            bf = bf + (1,)
        if len(bf) == 3 and bf[2] != 0:
            raise NotImplementedError('basis functions in different spaces')
        out.append((str(bf[0]), int(bf[1])))
    return out


def evaluate(expr, G, X, inputs, bfuns=None, normal=None):
    """Evaluate a form string on a grid.

    G: grid shape; X: physical coordinates G + (d,); bfuns: [(name, components)], trial function first (arity 2) or the
    test function alone (arity 1); normal: outer unit normals G + (d,) for boundary integrals.
    Returns (arity, measure, table, ncs): arity 2 -> table[a][b][r][s] (a: test component, r: its jet index; b, s: trial)
    of arrays over G or None; arity 1 -> table[a][r]; ncs = components of (trial, test) or (test,)."""
    d = X.shape[-1]
    bf = normalise_bfuns(expr, bfuns)
    if len(bf) not in (1, 2):
        raise ValueError('arity should be 1 or 2')
    J = d + 1
    ns = {}
    for name, nc in bf:
        who = (name, name == bf[-1][0])                   # the last function of `bfuns` is the test function
        a = np.zeros((1,) * len(G) + ((nc,) if nc > 1 else ()) + (nc, J))
        if nc > 1:
            for c in range(nc):
                a[(0,) * len(G) + (c, c, 0)] = 1.0
            ns[name] = TExpr(a, (nc,), 'l', who, True)
        else:
            a[..., 0, 0] = 1.0
            ns[name] = TExpr(a, (), 'l', who, True)
    grad = make_grad(d)

    def div(e):
        m = grad(e)
        if len(m.shape) != 2 or m.shape[0] != m.shape[1]:
            raise NotImplementedError('div() of a non-vector (or of a vector whose length is not the space dimension)')
        return tr(m)

    def curl(e):
        m = grad(e)                                   # m[i][k] = d_k e_i
        if m.shape == (3, 3):
            return stack([m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1]])
        if m.shape == (2, 2):
            return m[1, 0] - m[0, 1]
        raise NotImplementedError('curl() of this shape')

    def coef_fn(f):
        return lambda c: TExpr(f(TExpr.wrap(c).a), TExpr.wrap(c).shape, 'c')

    ns.update({'grad': grad, 'div': div, 'curl': curl, 'inner': inner, 'dot': dot, 'cross': cross, 'outer': outer, 'tr': tr,
               'dx': _Measure('dx'), 'ds': _Measure('ds'), 'x': TExpr(X, (d,), 'c'),
               'as_vector': lambda c: stack(list(c)), 'as_matrix': lambda m: stack([stack(list(r)) for r in m]),
               'sqrt': coef_fn(np.sqrt), 'exp': coef_fn(np.exp), 'sin': coef_fn(np.sin), 'cos': coef_fn(np.cos)})
    if normal is not None:
        ns['n'] = TExpr(normal, (d,), 'c')
    for name, val in inputs.items():
        if name == 'geo' or name in ns and name in dict(bf):
            continue
        if callable(val):
            vals = val(*(X[..., k] for k in range(d)))
            if isinstance(vals, (tuple, list)):
                ns[name] = stack([_field(v, G) for v in vals])
                continue
            ns[name] = _field(vals, G)
        else:
            a = np.asarray(val, dtype=float)
            ns[name] = TExpr(a, a.shape, 'c', const=True) if a.shape[:len(G)] != G or not len(G) else _field(a, G)
    try:
        res = eval(expr, {'__builtins__': {}}, ns)
    except NameError as e:
        raise ValueError('unknown name in the form: %s' % e)
    if not isinstance(res, Measured):
        raise NotImplementedError('the form must be an integral (... * dx or ... * ds)')
    e = res.e
    ncs = tuple(nc for _, nc in bf)
    if len(bf) == 2:
        if e.kind != 'b':
            raise NotImplementedError('the form must be bilinear in %s and %s' % (bf[0][0], bf[1][0]))
        a = np.broadcast_to(e.a, G + e.a.shape[len(G):]) if e.a.shape[:len(G)] != G else e.a
        ncv, ncu = ncs[1], ncs[0]
        table = [[[[_nz(a[..., p, r, q, s]) for s in range(J)] for r in range(J)] for q in range(ncu)] for p in range(ncv)]
        return 2, res.measure, table, ncs
    if e.kind != 'l':
        raise NotImplementedError('the form must be linear in %s' % bf[0][0])
    a = np.broadcast_to(e.a, G + e.a.shape[len(G):]) if e.a.shape[:len(G)] != G else e.a
    table = [[_nz(a[..., p, r]) for r in range(J)] for p in range(ncs[0])]
    return 1, res.measure, table, ncs


def _field(vals, G):
    vals = np.asarray(vals, dtype=float)
    if vals.shape[:len(G)] == G:
        return TExpr(vals, vals.shape[len(G):], 'c')
    # a function that ignores some of its arguments returns fewer grid axes: broadcast (pyiga/utils.py:17-31)
    return TExpr(np.broadcast_to(vals, G), (), 'c') if vals.ndim <= len(G) else TExpr(vals, vals.shape, 'c', const=True)


def _nz(a):
    return None if not np.any(a != 0.0) else np.ascontiguousarray(a, dtype=float)
