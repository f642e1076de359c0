"""Front-end for variational-form strings (the `problem` argument of ``assemble.assemble``).

The reference evaluates the string with its symbolic ``VForm`` objects and compiles the result to
Cython (pyiga/vform.py:1804-1885 ``parse_vf``, pyiga/codegen/cython.py, pyiga/compile.py).  Here the
string is evaluated with the small symbolic classes below, directly on the Gauss grid: every
coefficient is a numpy array over the grid, and the outcome is the table of physical coefficient
fields ``P[r][s]`` of

    a(u, v) = integral of  sum_{r,s=0..3}  P_rs(x) * D_r v * D_s u  dx,      D_0 = id, D_1..3 = d/dx, d/dy, d/dz

which is what ``IGX_FORM`` of libigx assembles (include/igx.h).  Supported: scalar trial/test functions
``u``, ``v``; ``grad``, ``inner``, ``dot``, ``dx``; ``+ - *`` and division by coefficients; numbers;
``x`` (physical coordinates, ``x[0..2]``); tuples as vector literals; named inputs (callables of the
physical coordinates returning scalars, tuples/vectors or ``(3,3)`` arrays; constants and arrays).
Anything else (higher derivatives, vector-valued bases, surface integrals, ``div``/``curl`` ...) raises
``NotImplementedError`` -- the general compiler is row f1 "full" of SURVEY section 8.
"""
import numpy as np

from . import symbolic


def _arr(x):
    """float array -- or the object array of a traced evaluation (pyiga_amd.symbolic), left as it is"""
    a = np.asarray(x)
    return a if a.dtype == object else a.astype(float)


def _uniform_value(a):
    """The value of an array that is the same at every grid point, else None (traced arrays: constants only)."""
    a = np.asarray(a)
    if a.dtype == object:
        consts = [getattr(z, 'const', z) for z in a.reshape(-1)]
        if any(c is None for c in consts) or any(c != consts[0] for c in consts):
            return None
        return float(consts[0])
    if a.ndim == 0 or np.all(a == a.flat[0]):
        return float(a.flat[0]) if a.ndim else float(a)
    return None


class _Coef:
    """A coefficient field on the grid: scalar (shape G), vector (G + (3,)) or matrix (G + (3, 3))."""
    __array_priority__ = 1000

    def __init__(self, a, rank):
        self.a, self.rank = a, rank

    @staticmethod
    def wrap(x, G):
        if isinstance(x, _Coef):
            return x
        if isinstance(x, (tuple, list)):
            comps = [_Coef.wrap(c, G) for c in x]
            if all(c.rank == 0 for c in comps):
                return _Coef(np.stack([np.broadcast_to(c.a, G) for c in comps], axis=-1), 1)
            if all(c.rank == 1 for c in comps):
                return _Coef(np.stack([c.a for c in comps], axis=-2), 2)
            raise NotImplementedError('unsupported nested tuple in the form')
        a = _arr(x)
        if a.ndim == 0:
            return _Coef(a, 0)
        d = len(G)
        if d and a.shape == (d,):
            return _Coef(np.broadcast_to(a, G + (d,)), 1)
        if d and a.shape == (d, d):
            return _Coef(np.broadcast_to(a, G + (d, d)), 2)
        rank = a.ndim - len(G)
        assert a.shape[:len(G)] == G and 0 <= rank <= 2, 'coefficient does not fit the Gauss grid'
        return _Coef(a, rank)

    def _ex(self, rank):            # array broadcastable against a coefficient of the given rank
        a = self.a
        return a if a.ndim == 0 else a.reshape(a.shape + (1,) * (rank - self.rank))

    def __getitem__(self, k):
        assert self.rank >= 1
        return _Coef(self.a[..., k] if self.rank == 1 else self.a[..., k, :], self.rank - 1)

    def _bin(self, other, op):
        if isinstance(other, (_Lin, _Bil, _Dx)) or getattr(other, '_is_form_expr', False):
            return NotImplemented           # (expressions in u, v -- also those of pyiga_amd.pforms -- handle the product)
        o = _Coef.wrap(other, ())
        r = max(self.rank, o.rank)
        if self.rank and o.rank and self.rank != o.rank:
            raise NotImplementedError('elementwise operation between coefficients of different rank')
        return _Coef(op(self._ex(r) if self.rank < r else self.a, o._ex(r) if o.rank < r else o.a), r)

    def __add__(self, o): return self._bin(o, np.add)
    __radd__ = __add__
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return _Coef.wrap(o, ())._bin(self, np.subtract)
    def __mul__(self, o): return self._bin(o, np.multiply)
    __rmul__ = __mul__
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return _Coef.wrap(o, ())._bin(self, np.divide)
    def __neg__(self): return _Coef(-self.a, self.rank)
    def __pos__(self): return self
    def __pow__(self, k): return _Coef(self.a ** k, self.rank)


class _Lin:
    """Expression that is linear in the jet of ONE basis function: scalar-valued  s*phi + w.grad(phi)
    (vector=False) or vector-valued  M grad(phi) + t*phi  (vector=True)."""
    __array_priority__ = 1000

    def __init__(self, who, vector, s=None, w=None, M=None, t=None):
        self.who, self.vector = who, vector
        self.s, self.w, self.M, self.t = s, w, M, t      # numpy arrays or None (= zero)

    def _scaled(self, c):
        c = c if isinstance(c, _Coef) else _Coef.wrap(c, ())
        if c.rank != 0:
            raise NotImplementedError('only scalar coefficients can multiply a basis-function expression; use dot() or inner()')
        f = lambda z, extra: None if z is None else z * (c.a if c.a.ndim == 0 else c.a.reshape(c.a.shape + (1,) * extra))
        return _Lin(self.who, self.vector, f(self.s, 0), f(self.w, 1), f(self.M, 2), f(self.t, 1))

    def __mul__(self, o):
        if isinstance(o, _Dx):
            return o.__rmul__(self)
        if isinstance(o, _Lin):
            return _product(self, o)
        if isinstance(o, _Bil):
            raise NotImplementedError('the form is not bilinear')
        return self._scaled(o)
    __rmul__ = __mul__

    def __truediv__(self, o): return self._scaled(1.0 / _Coef.wrap(o, ()))
    def __neg__(self): return self._scaled(-1.0)
    def __pos__(self): return self

    def __add__(self, o):
        if not (isinstance(o, _Lin) and o.who == self.who and o.vector == self.vector):
            raise NotImplementedError('sum of incompatible expressions')
        ad = lambda a, b: b if a is None else (a if b is None else a + b)
        return _Lin(self.who, self.vector, ad(self.s, o.s), ad(self.w, o.w), ad(self.M, o.M), ad(self.t, o.t))

    def __sub__(self, o): return self + (-o)


class _Bil:
    """Table of physical coefficients P[r][s] (r: jet index of v, s: of u); entries are arrays or None."""
    __array_priority__ = 1000

    def __init__(self, P=None, measured=False, n=4):
        self.P = P if P is not None else [[None] * n for _ in range(n)]
        self.measured = measured

    @property
    def n(self):
        return len(self.P)

    def _map(self, f):
        return _Bil([[None if e is None else f(e) for e in row] for row in self.P], self.measured)

    def __mul__(self, o):
        if isinstance(o, _Dx):
            assert not self.measured, 'dx applied twice'
            return _Bil(self.P, True)
        if isinstance(o, (_Lin, _Bil)):
            raise NotImplementedError('the form is not bilinear')
        c = _Coef.wrap(o, ())
        if c.rank != 0:
            raise NotImplementedError('only scalar coefficients can multiply an integrand')
        return self._map(lambda e: e * c.a)
    __rmul__ = __mul__

    def __truediv__(self, o): return self * (1.0 / _Coef.wrap(o, ()))
    def __neg__(self): return self._map(lambda e: -e)
    def __pos__(self): return self

    def __add__(self, o):
        if not isinstance(o, _Bil) or o.measured != self.measured:
            raise NotImplementedError('sum of incompatible expressions (is every term multiplied by dx?)')
        ad = lambda a, b: b if a is None else (a if b is None else a + b)
        return _Bil([[ad(self.P[r][s], o.P[r][s]) for s in range(self.n)] for r in range(self.n)], self.measured)

    def __sub__(self, o): return self + (-o)


class _Functional:
    """Integrand of a linear functional: scalar-valued expression in the jet of v, times dx."""
    def __init__(self, lin):
        self.lin = lin


class _Dx:
    def __call__(self, *a, **k):
        raise NotImplementedError('partial derivatives Dx()/dx() of basis functions are not supported: use grad()')

    def __rmul__(self, o):
        if isinstance(o, _Bil):
            return o * self
        if isinstance(o, _Lin) and not o.vector:
            return _Functional(o)
        raise NotImplementedError('dx must multiply a scalar integrand in u and/or v')
    __mul__ = __rmul__


def _jet(lin):
    """Scalar-valued linear expression -> list of 4 coefficient arrays (value, d/dx, d/dy, d/dz) or None."""
    assert not lin.vector
    d = (lin.s if lin.w is None else lin.w[..., 0]).ndim
    return [lin.s] + [None if lin.w is None else lin.w[..., k] for k in range(d)]


def _product(a, b):
    if a.who == b.who:
        raise NotImplementedError('the form is not bilinear in (u, v)')
    if a.vector or b.vector:
        raise NotImplementedError('product of vector-valued expressions: use inner()')
    u, v = (a, b) if a.who == 'u' else (b, a)
    ju, jv = _jet(u), _jet(v)
    n = len(ju)
    B = _Bil(n=n)
    for r in range(n):
        for s in range(n):
            if jv[r] is not None and ju[s] is not None:
                B.P[r][s] = jv[r] * ju[s]
    return B


def _vec_jet(lin):
    """Vector-valued expression  M grad(phi) + t phi  -> list over components c of the 4 jet coefficients."""
    out = []
    d = (lin.M.shape[-1] if lin.M is not None else lin.t.shape[-1])
    for c in range(d):
        s = None if lin.t is None else lin.t[..., c]
        w = [None if lin.M is None else lin.M[..., c, k] for k in range(d)]
        out.append([s] + w)
    return out


def make_namespace(G, X, inputs):
    """Names available to a form string.  G: grid shape; X: physical coordinates, G + (3,); inputs: dict of
    callables (evaluated at the physical coordinates) or constants."""
    d = len(G)
    assert X.shape == G + (d,), 'physical coordinates do not fit the grid'
    one = np.ones(G)
    eye = np.broadcast_to(np.eye(d), G + (d, d))

    def basis(who):
        return _Lin(who, False, s=one)

    def higher(name):
        def f(*a, **k):
            raise NotImplementedError('%s(): second / parametric derivatives are the business of pyiga_amd.pforms' % name)
        return f

    def grad(e, dims=None, parametric=False):
        if dims is not None or parametric:
            raise NotImplementedError('grad(dims=..., parametric=...): see pyiga_amd.pforms')
        if not (isinstance(e, _Lin) and not e.vector and e.w is None):
            raise NotImplementedError('grad() of anything but u or v (times a constant)')
        c = _uniform_value(e.s)
        if c is None:
            raise NotImplementedError('grad() of a basis function times a variable coefficient')
        return _Lin(e.who, True, M=eye * c)

    def inner(a, b):
        if isinstance(a, _Lin) and isinstance(b, _Lin):
            if a.who == b.who:
                raise NotImplementedError('the form is not bilinear in (u, v)')
            if not (a.vector and b.vector):
                raise NotImplementedError('inner() of scalar expressions: use *')
            u, v = (a, b) if a.who == 'u' else (b, a)
            ju, jv = _vec_jet(u), _vec_jet(v)
            B = _Bil(n=d + 1)
            for c in range(d):
                for r in range(d + 1):
                    for s in range(d + 1):
                        if jv[c][r] is not None and ju[c][s] is not None:
                            term = jv[c][r] * ju[c][s]
                            B.P[r][s] = term if B.P[r][s] is None else B.P[r][s] + term
            return B
        if isinstance(b, _Lin):
            a, b = b, a
        if isinstance(a, _Lin):         # inner(vector expression, coefficient vector) -> scalar-valued expression
            if not a.vector:
                raise NotImplementedError('inner() of a scalar expression')
            c = _Coef.wrap(b, G)
            if c.rank != 1:
                raise NotImplementedError('inner() of a gradient with a non-vector coefficient')
            s = None if a.t is None else np.einsum('...c,...c->...', a.t, c.a)
            w = None if a.M is None else np.einsum('...ck,...c->...k', a.M, c.a)
            return _Lin(a.who, False, s=s, w=w)
        ca, cb = _Coef.wrap(a, G), _Coef.wrap(b, G)
        return _Coef(np.sum(ca.a * cb.a, axis=tuple(range(-ca.rank, 0)) if ca.rank else None), 0)

    def dot(a, b):
        if isinstance(b, _Lin) and not isinstance(a, _Lin):      # matrix (or scalar) times vector expression
            K = _Coef.wrap(a, G)
            if K.rank == 0:
                return b * K
            if K.rank != 2 or not b.vector:
                raise NotImplementedError('dot(): expected a matrix coefficient and a gradient')
            M = None if b.M is None else np.einsum('...cd,...dk->...ck', K.a, b.M)
            t = None if b.t is None else np.einsum('...cd,...d->...c', K.a, b.t)
            return _Lin(b.who, True, M=M, t=t)
        if isinstance(a, _Lin) and not isinstance(b, _Lin):
            K = _Coef.wrap(b, G)
            if K.rank == 1:
                return inner(a, K)
            raise NotImplementedError('dot(gradient, matrix)')
        if isinstance(a, _Lin):
            return inner(a, b)
        ca, cb = _Coef.wrap(a, G), _Coef.wrap(b, G)
        if ca.rank == 2 and cb.rank == 1:
            return _Coef(np.einsum('...cd,...d->...c', ca.a, cb.a), 1)
        if ca.rank == 1 and cb.rank == 1:
            return inner(ca, cb)
        if ca.rank == 2 and cb.rank == 2:
            return _Coef(np.einsum('...cd,...de->...ce', ca.a, cb.a), 2)
        raise NotImplementedError('dot() of these coefficient shapes')

    ns = {'u': basis('u'), 'v': basis('v'), 'grad': grad, 'inner': inner, 'dot': dot, 'dx': _Dx(),
          'hess': higher('hess'), 'Dx': higher('Dx'), 'div': higher('div'), 'tr': higher('tr'),
          'x': _Coef(X, 1),
          'sqrt': lambda c: _Coef(np.sqrt(_Coef.wrap(c, G).a), _Coef.wrap(c, G).rank),
          'exp': lambda c: _Coef(np.exp(_Coef.wrap(c, G).a), _Coef.wrap(c, G).rank),
          'as_vector': lambda c: _Coef.wrap(tuple(c), G), 'as_matrix': lambda c: _Coef.wrap(tuple(tuple(r) for r in c), G)}
    for name, val in inputs.items():
        if name in ('geo',):
            continue
        if hasattr(val, 'grid_eval') and not callable(val):
            raise NotImplementedError('spline functions as form inputs')
        if callable(val):
            vals = val(*(X[..., k] for k in range(d)))
            if not isinstance(vals, (tuple, list)):
                # a function that ignores some of its arguments returns fewer grid axes: broadcast like the
                # reference does (pyiga/utils.py:17-31)
                vals = _arr(vals)
                extra = vals.shape[len(G):] if vals.shape[:len(G)] == G else (vals.shape if vals.shape in ((d,), (d, d)) else ())
                vals = np.broadcast_to(vals, G + extra)
            ns[name] = _Coef.wrap(vals, G)
        else:
            ns[name] = _Coef.wrap(val, G)
    return ns


def coefficient_table(expr, G, X, inputs, traced=False):
    """Evaluate the form string; returns the 4x4 table of coefficient arrays (shape G) or None.  traced: the arrays are
    object arrays of expression trees (symbolic_table)."""
    ns = make_namespace(G, X, inputs)
    try:
        res = eval(expr, {'__builtins__': {}}, ns)
    except NameError as e:
        raise ValueError('unknown name in the form: %s' % e)
    if not isinstance(res, _Bil) or not res.measured:
        raise NotImplementedError('the form must be a volume integral (... * dx) that is bilinear in u and v')
    n = len(G) + 1
    table = [[None] * n for _ in range(n)]
    for r in range(n):
        for s in range(n):
            e = res.P[r][s]
            if e is not None and symbolic.any_nonzero(e):
                table[r][s] = np.broadcast_to(e, G) if traced else np.ascontiguousarray(np.broadcast_to(e, G), dtype=float)
    return table


def symbolic_table(expr, d, inputs):
    """The coefficient table of the form string as C expressions in the physical coordinates (x, y[, z]): the string -- and
    every callable among the inputs -- is evaluated ONCE, on a grid of one point whose coordinates are symbols
    (pyiga_amd.symbolic).  Raises (symbolic.NotTraceable or whatever numpy makes of the attempt) when something cannot be
    traced; the caller then samples the coefficients on the host."""
    from . import symbolic
    table = coefficient_table(expr, (1,) * d, symbolic.coordinates(d), inputs, traced=True)
    return [[None if e is None else symbolic.c_source(e) for e in row] for row in table]


def functional_jet(expr, G, X, inputs, traced=False):
    """Evaluate an arity-1 form string ``(F0 * v + inner(F, grad(v))) * dx``; returns the list
    ``[F0, F_1, ..., F_d]`` of coefficient arrays on the grid (shape G) or None."""
    ns = make_namespace(G, X, inputs)
    ns.pop('u')
    try:
        res = eval(expr, {'__builtins__': {}}, ns)
    except NameError as e:
        raise ValueError('unknown name in the form: %s' % e)
    if not isinstance(res, _Functional) or res.lin.who != 'v':
        raise NotImplementedError('the form must be a volume integral (... * dx) that is linear in v')
    out = []
    for e in _jet(res.lin):
        if e is None or not symbolic.any_nonzero(e):
            out.append(None)
        else:
            out.append(np.broadcast_to(e, G) if traced else np.ascontiguousarray(np.broadcast_to(e, G), dtype=float))
    return out


def functional_coefficient(expr, G, X, inputs):
    """Value coefficient of an arity-1 form string ``F * v * dx`` (no derivatives of v allowed)."""
    jet = functional_jet(expr, G, X, inputs)
    if any(e is not None for e in jet[1:]):
        raise NotImplementedError('this functional contains derivatives of v: use functional_jet')
    return jet[0] if jet[0] is not None else np.zeros(G)


def arity(expr):
    """1 if only v occurs in the form string, 2 if u and v do (as the reference decides: pyiga/vform.py:1818-1825)."""
    import re
    words = set(re.findall(r"[^\d\W]\w*", expr))
    used = words & {'u', 'v'}
    if used == {'v'}:
        return 1
    if used == {'u', 'v'}:
        return 2
    if used == {'u'}:
        raise NotImplementedError('a form that contains u but not v: name the test function v')
    raise ValueError('arity should be 1 or 2')
