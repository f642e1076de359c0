"""Form strings with second derivatives and parametric derivatives (``hess``, ``Dx(e, k, times=2)``, ``div(grad(u))``,
``grad(u, parametric=True)`` ...): the part of the reference's form language that ``pyiga_amd.forms`` (first-order jets,
transformed on the device) does not cover.

The reference keeps derivative multi-indices symbolically and lets its code generator transform physical derivatives to
parametric ones -- first derivatives with the inverse Jacobian, second derivatives with the inverse Jacobian and the
Hessian of the geometry map (pyiga/vform.py:540-625; Dx / grad / hess / div: pyiga/vform.py:1518-1600).  Here the same
transformation is applied to coefficient ARRAYS on the Gauss grid, and the outcome is a parametric jet form

    a(u, v) = sum_k  integral of  c_k(xi) * D^(ov_k) v * D^(ou_k) u  d xi,        ov_k, ou_k: derivative orders 0..2 per axis,

which the device assembles in passes (``igx_patch_set_pform`` + ``igx_patch_set_basis_orders`` of libigx, include/igx.h):
the two slots of an axis' basis table hold two derivative orders, and a pass takes the terms whose orders fit one choice
of slots per axis (``plan_passes``).

Supported: scalar trial / test functions ``u``, ``v``; ``Dx``, ``grad``, ``hess``, ``div`` (each physical or
``parametric=True``) applied to a basis function (times a constant) up to total order 2; ``inner``, ``dot``, ``tr``,
``as_vector``, ``as_matrix``, indexing; ``+ - * /`` with coefficients; ``x``; named inputs as in ``pyiga_amd.forms``; ``dx``.
"""
import itertools

import numpy as np

from .forms import _Coef

MAXORDER = 2


def _const_value(c):
    """The value of a coefficient that is the same at every grid point, else None."""
    a = np.asarray(c, dtype=float)
    if a.ndim == 0:
        return float(a)
    return float(a.flat[0]) if np.all(a == a.flat[0]) else None


def _scalar_coef(o, what):
    c = o if isinstance(o, _Coef) else _Coef.wrap(o, ())
    if c.rank != 0:
        raise NotImplementedError('only scalar coefficients can multiply %s; use dot() or inner()' % what)
    return c.a


class _Geo:
    """Geometry data on the Gauss grid: Jac[..., i, k] = d x_i / d xi_k and H2[..., m, e, u] = d^2 x_m / d xi_e d xi_u, all
    indices in (x, y, z) order."""

    def __init__(self, G, Jac, H2):
        self.G, self.d = tuple(G), len(G)
        self.Jac, self.H2 = np.asarray(Jac, dtype=float), None if H2 is None else np.asarray(H2, dtype=float)
        assert self.Jac.shape == self.G + (self.d, self.d), 'Jacobians do not fit the Gauss grid'
        self.JI = np.linalg.inv(self.Jac)                   # JI[..., k, i] = d xi_k / d x_i
        self.absdet = np.abs(np.linalg.det(self.Jac))
        self._gamma = None

    @property
    def gamma(self):
        """Gamma[..., k, i, j] = d^2 xi_k / d x_i d x_j = - sum_meu H2[m, e, u] JI[k, m] JI[e, i] JI[u, j]
        (pyiga/vform.py:609-625)."""
        if self._gamma is None:
            assert self.H2 is not None, 'second physical derivatives need the Hessian of the geometry map'
            self._gamma = -np.einsum('...meu,...km,...ei,...uj->...kij', self.H2, self.JI, self.JI, self.JI)
        return self._gamma


def _unit(d, *ks):
    a = [0] * d
    for k in ks:
        a[k] += 1
    return tuple(a)


def _acc(table, key, val):
    table[key] = val if key not in table else table[key] + val


class _Jet:
    """Scalar-valued expression that is linear in the derivatives of ONE basis function:  sum c * D^alpha phi  with
    `terms` = {(alpha, parametric): c}, alpha a derivative multi-index in (x, y, z) order (physical or parametric
    coordinates), c a number or an array on the grid."""
    __array_priority__ = 1000
    __array_ufunc__ = None
    _is_form_expr = True

    def __init__(self, who, terms):
        self.who, self.terms = who, terms

    def _scaled(self, c):
        return _Jet(self.who, {k: v * c for k, v in self.terms.items()})

    def __mul__(self, o):
        if isinstance(o, _Measure):
            raise NotImplementedError('dx must multiply an integrand that is bilinear in u and v')
        if isinstance(o, _Jet):
            return _product(self, o)
        if isinstance(o, (_Bilinear, _JetTensor)):
            return NotImplemented
        return self._scaled(_scalar_coef(o, 'a basis-function expression'))
    __rmul__ = __mul__

    def __truediv__(self, o): return self._scaled(1.0 / _scalar_coef(o, 'a basis-function expression'))
    def __neg__(self): return self._scaled(-1.0)
    def __pos__(self): return self

    def __add__(self, o):
        if not (isinstance(o, _Jet) and o.who == self.who):
            raise NotImplementedError('sum of incompatible expressions')
        t = dict(self.terms)
        for k, v in o.terms.items():
            _acc(t, k, v)
        return _Jet(self.who, t)
    __radd__ = __add__

    def __sub__(self, o): return self + (-o)

    def derivative(self, k, times, parametric):
        out = {}
        for (alpha, par), c in self.terms.items():
            if _const_value(c) is None:
                raise NotImplementedError('derivative of a basis function times a variable coefficient')
            if any(alpha) and par != bool(parametric):
                raise NotImplementedError('physical derivative of a parametric derivative (or the other way round)')
            beta = list(alpha)
            beta[k] += times
            if sum(beta) > MAXORDER:
                raise NotImplementedError('derivatives of order %d of a basis function' % sum(beta))
            _acc(out, (tuple(beta), bool(parametric)), c)
        return _Jet(self.who, out)

    def parametric_terms(self, geo):
        """{alpha: coefficient} with PARAMETRIC multi-indices only."""
        d, out = geo.d, {}
        for (alpha, par), c in self.terms.items():
            order = sum(alpha)
            if par or order == 0:
                _acc(out, alpha, c)
                continue
            idx = [k for k in range(d) for _ in range(alpha[k])]       # the physical directions, ascending
            if order == 1:
                (i,) = idx
                for k in range(d):
                    _acc(out, _unit(d, k), c * geo.JI[..., k, i])
            else:
                i, j = idx
                for k in range(d):
                    for l in range(d):
                        _acc(out, _unit(d, k, l), c * (geo.JI[..., k, i] * geo.JI[..., l, j]))
                    _acc(out, _unit(d, k), c * geo.gamma[..., k, i, j])
        return out


class _JetTensor:
    """Vector or matrix of _Jet expressions (gradients, Hessians and what dot() makes of them)."""
    __array_priority__ = 1000
    __array_ufunc__ = None
    _is_form_expr = True

    def __init__(self, e):
        self.e = np.asarray(e, dtype=object)

    @property
    def shape(self):
        return self.e.shape

    def __getitem__(self, k):
        r = self.e[k]
        return _JetTensor(r) if isinstance(r, np.ndarray) else r

    def _map(self, f):
        out = np.empty(self.e.shape, dtype=object)
        for idx in np.ndindex(*self.e.shape):
            out[idx] = f(self.e[idx])
        return _JetTensor(out)

    def __mul__(self, o):
        if isinstance(o, (_Jet, _JetTensor, _Bilinear, _Measure)):
            raise NotImplementedError('product of a vector-valued expression: use inner() or dot()')
        c = _scalar_coef(o, 'a vector-valued expression')
        return self._map(lambda z: z * c)
    __rmul__ = __mul__

    def __truediv__(self, o):
        c = 1.0 / _scalar_coef(o, 'a vector-valued expression')
        return self._map(lambda z: z * c)

    def __neg__(self): return self._map(lambda z: -z)
    def __pos__(self): return self

    def __add__(self, o):
        if not (isinstance(o, _JetTensor) and o.shape == self.shape):
            raise NotImplementedError('sum of expressions of different shape')
        out = np.empty(self.e.shape, dtype=object)
        for idx in np.ndindex(*self.e.shape):
            out[idx] = self.e[idx] + o.e[idx]
        return _JetTensor(out)

    def __sub__(self, o): return self + (-o)

    @property
    def T(self):
        return _JetTensor(self.e.T)


class _Bilinear:
    """sum c * D^av v * D^au u with parametric multi-indices: {(av, au): c}; `measured` after ``* dx``."""
    __array_priority__ = 1000
    __array_ufunc__ = None
    _is_form_expr = True

    def __init__(self, terms, geo, measured=False):
        self.terms, self.geo, self.measured = terms, geo, measured

    def _scaled(self, c):
        return _Bilinear({k: v * c for k, v in self.terms.items()}, self.geo, self.measured)

    def __mul__(self, o):
        if isinstance(o, _Measure):
            assert not self.measured, 'dx applied twice'
            return _Bilinear({k: v * self.geo.absdet for k, v in self.terms.items()}, self.geo, True)
        if isinstance(o, (_Jet, _JetTensor, _Bilinear)):
            raise NotImplementedError('the form is not bilinear')
        return self._scaled(_scalar_coef(o, 'an integrand'))
    __rmul__ = __mul__

    def __truediv__(self, o): return self._scaled(1.0 / _scalar_coef(o, 'an integrand'))
    def __neg__(self): return self._scaled(-1.0)
    def __pos__(self): return self

    def __add__(self, o):
        if isinstance(o, (int, float)) and o == 0:                   # sum(...) starts from 0
            return self
        if not isinstance(o, _Bilinear) or o.measured != self.measured:
            raise NotImplementedError('sum of incompatible expressions (is every term multiplied by dx?)')
        t = dict(self.terms)
        for k, v in o.terms.items():
            _acc(t, k, v)
        return _Bilinear(t, self.geo, self.measured)
    __radd__ = __add__

    def __sub__(self, o): return self + (-o)


class _Measure:
    _is_form_expr = True

    def __call__(self, *a, **k):
        raise NotImplementedError('dx() as a derivative: use Dx(e, k)')

    def __rmul__(self, o):
        if isinstance(o, _Bilinear):
            return o * self
        raise NotImplementedError('dx must multiply an integrand that is bilinear in u and v')
    __mul__ = __rmul__


_GEO = [None]       # geometry data of the form being evaluated (set by evaluate(); products of jets need it)


def _product(a, b):
    if a.who == b.who:
        raise NotImplementedError('the form is not bilinear in (u, v)')
    u, v = (a, b) if a.who == 'u' else (b, a)
    geo = _GEO[0]
    tu, tv = u.parametric_terms(geo), v.parametric_terms(geo)
    out = {}
    for av, cv in tv.items():
        for au, cu in tu.items():
            _acc(out, (av, au), cv * cu)
    return _Bilinear(out, geo)


def make_namespace(geo, X, inputs):
    """Names available to a form string; geo: _Geo, X: physical coordinates G + (d,)."""
    G, d = geo.G, geo.d
    zero = (0,) * d

    def basis(who):
        return _Jet(who, {(zero, False): 1.0})

    def Dx(e, k, times=1, parametric=False):
        if isinstance(e, _JetTensor):
            return e._map(lambda z: Dx(z, k, times, parametric))
        if not isinstance(e, _Jet):
            raise NotImplementedError('Dx() of anything but an expression in u or v')
        if not 0 <= k < d:
            raise ValueError('Dx(): coordinate %d of a %dD patch' % (k, d))
        return e.derivative(k, times, parametric)

    def grad(e, dims=None, parametric=False):
        dims = range(d) if dims is None else dims
        if isinstance(e, _Jet):
            return _JetTensor([Dx(e, k, 1, parametric) for k in dims])
        if isinstance(e, _JetTensor) and len(e.shape) == 1:
            return _JetTensor([[Dx(z, k, 1, parametric) for k in dims] for z in e.e])
        raise NotImplementedError('grad() of this expression')

    def hess(e, parametric=False):
        if not isinstance(e, _Jet):
            raise NotImplementedError('hess() of anything but a scalar expression in u or v')
        return grad(grad(e, parametric=parametric), parametric=parametric)

    def div(e, parametric=False):
        if not (isinstance(e, _JetTensor) and e.shape == (d,)):
            raise NotImplementedError('div() of anything but a vector expression in u or v')
        out = Dx(e.e[0], 0, 1, parametric)
        for k in range(1, d):
            out = out + Dx(e.e[k], k, 1, parametric)
        return out

    def tr(e):
        if isinstance(e, _JetTensor) and len(e.shape) == 2 and e.shape[0] == e.shape[1]:
            out = e.e[0, 0]
            for k in range(1, e.shape[0]):
                out = out + e.e[k, k]
            return out
        c = _Coef.wrap(e, G)
        if c.rank != 2:
            raise NotImplementedError('tr() of a non-matrix')
        return _Coef(np.trace(c.a, axis1=-2, axis2=-1), 0)

    def _coef_elems(c, shape):
        """Object array of the components of a coefficient tensor of the given shape."""
        c = _Coef.wrap(c, G)
        if c.rank != len(shape):
            raise NotImplementedError('coefficient of rank %d against an expression of shape %s' % (c.rank, shape))
        a = np.broadcast_to(c.a, G + tuple(shape)) if c.a.ndim >= len(shape) and c.a.shape[-len(shape):] == tuple(shape) else None
        if a is None:
            raise NotImplementedError('coefficient shape does not match the expression')
        out = np.empty(shape, dtype=object)
        for idx in np.ndindex(*shape):
            out[idx] = a[(Ellipsis,) + idx]
        return out

    def inner(a, b):
        if isinstance(a, _JetTensor) and isinstance(b, _JetTensor):
            if a.shape != b.shape:
                raise NotImplementedError('inner() of expressions of different shape')
            out = 0
            for idx in np.ndindex(*a.shape):
                out = a.e[idx] * b.e[idx] + out
            return out
        if isinstance(a, _Jet) and isinstance(b, _Jet):
            return a * b
        if isinstance(b, _JetTensor):
            a, b = b, a
        if isinstance(a, _JetTensor):
            ce = _coef_elems(b, a.shape)
            out = None
            for idx in np.ndindex(*a.shape):
                t = a.e[idx] * _Coef(ce[idx], 0)
                out = t if out is None else out + t
            return out
        if isinstance(a, _Jet) or isinstance(b, _Jet):
            raise NotImplementedError('inner() of a scalar expression with a coefficient: use *')
        ca, cb = _Coef.wrap(a, G), _Coef.wrap(b, G)
        return _Coef(np.sum(ca.a * cb.a, axis=tuple(range(-ca.rank, 0)) if ca.rank else None), 0)

    def dot(a, b):
        ja, jb = isinstance(a, _JetTensor), isinstance(b, _JetTensor)
        if ja and jb:
            if len(a.shape) == 1 and len(b.shape) == 1:
                return inner(a, b)
            raise NotImplementedError('dot() of two matrix-valued expressions in u and v')
        if jb:                                      # coefficient (matrix / vector / scalar) . expression
            K = _Coef.wrap(a, G)
            if K.rank == 0:
                return b * K
            if K.rank == 2:
                Ke = _coef_elems(K, K.a.shape[-2:])
                if Ke.shape[1] != b.shape[0]:
                    raise NotImplementedError('dot(): shapes do not match')
                rows = []
                for i in range(Ke.shape[0]):
                    if len(b.shape) == 1:
                        r = None
                        for j in range(Ke.shape[1]):
                            t = b.e[j] * _Coef(Ke[i, j], 0)
                            r = t if r is None else r + t
                        rows.append(r)
                    else:
                        row = []
                        for c in range(b.shape[1]):
                            r = None
                            for j in range(Ke.shape[1]):
                                t = b.e[j, c] * _Coef(Ke[i, j], 0)
                                r = t if r is None else r + t
                            row.append(r)
                        rows.append(row)
                return _JetTensor(rows)
            if K.rank == 1 and len(b.shape) == 1:
                return inner(b, K)
            if K.rank == 1 and len(b.shape) == 2:   # row vector . matrix
                ce = _coef_elems(K, (b.shape[0],))
                cols = []
                for c in range(b.shape[1]):
                    r = None
                    for j in range(b.shape[0]):
                        t = b.e[j, c] * _Coef(ce[j], 0)
                        r = t if r is None else r + t
                    cols.append(r)
                return _JetTensor(cols)
            raise NotImplementedError('dot() of these shapes')
        if ja:                                      # expression . coefficient
            K = _Coef.wrap(b, G)
            if K.rank == 0:
                return a * K
            if K.rank == 1 and len(a.shape) == 1:
                return inner(a, K)
            if K.rank == 1 and len(a.shape) == 2:   # matrix . vector
                ce = _coef_elems(K, (a.shape[1],))
                rows = []
                for i in range(a.shape[0]):
                    r = None
                    for j in range(a.shape[1]):
                        t = a.e[i, j] * _Coef(ce[j], 0)
                        r = t if r is None else r + t
                    rows.append(r)
                return _JetTensor(rows)
            if K.rank == 2 and len(a.shape) == 1:   # row vector . matrix
                return dot(_Coef(np.swapaxes(K.a, -1, -2), 2), a)
            raise NotImplementedError('dot() of these shapes')
        ca, cb = _Coef.wrap(a, G), _Coef.wrap(b, G)
        if ca.rank == 2 and cb.rank == 1:
            return _Coef(np.einsum('...cd,...d->...c', ca.a, cb.a), 1)
        if ca.rank == 1 and cb.rank == 1:
            return inner(ca, cb)
        if ca.rank == 2 and cb.rank == 2:
            return _Coef(np.einsum('...cd,...de->...ce', ca.a, cb.a), 2)
        raise NotImplementedError('dot() of these coefficient shapes')

    def as_vector(c):
        c = tuple(c)
        if any(isinstance(z, _Jet) for z in c):
            return _JetTensor(list(c))
        return _Coef.wrap(c, G)

    def as_matrix(c):
        rows = tuple(tuple(r) for r in c)
        if any(isinstance(z, _Jet) for r in rows for z in r):
            return _JetTensor([list(r) for r in rows])
        return _Coef.wrap(rows, G)

    ns = {'u': basis('u'), 'v': basis('v'), 'Dx': Dx, 'grad': grad, 'hess': hess, 'div': div, 'tr': tr,
          'inner': inner, 'dot': dot, 'dx': _Measure(), 'x': _Coef(X, 1), 'as_vector': as_vector, 'as_matrix': as_matrix,
          'sqrt': lambda c: _Coef(np.sqrt(_Coef.wrap(c, G).a), _Coef.wrap(c, G).rank),
          'exp': lambda c: _Coef(np.exp(_Coef.wrap(c, G).a), _Coef.wrap(c, G).rank)}
    for name, val in inputs.items():
        if name in ('geo',):
            continue
        if hasattr(val, 'grid_eval') and not callable(val):
            raise NotImplementedError('spline functions as form inputs')
        if callable(val):
            vals = val(*(X[..., k] for k in range(d)))
            if not isinstance(vals, (tuple, list)):
                vals = np.asarray(vals, dtype=float)
                extra = vals.shape[len(G):] if vals.shape[:len(G)] == G else (vals.shape if vals.shape in ((d,), (d, d)) else ())
                vals = np.broadcast_to(vals, G + extra)
            ns[name] = _Coef.wrap(vals, G)
        else:
            ns[name] = _Coef.wrap(val, G)
    return ns


def evaluate(expr, G, X, Jac, H2, inputs):
    """Evaluate the form string on the Gauss grid.  Returns the list of terms ``(ov, ou, c)``: derivative orders of v and of u
    per GRID axis (tuples; x is the last grid axis) and the coefficient array (shape G) of  c * D^ov v * D^ou u  integrated over
    the parameter domain (|det J| is inside c, the Gauss weights are not)."""
    geo = _Geo(G, Jac, H2)
    ns = make_namespace(geo, X, inputs)
    _GEO[0] = geo
    try:
        res = eval(expr, {'__builtins__': {}}, ns)
    except NameError as e:
        raise ValueError('unknown name in the form: %s' % e)
    finally:
        _GEO[0] = None
    if not isinstance(res, _Bilinear) or not res.measured:
        raise NotImplementedError('the form must be a volume integral (... * dx) that is bilinear in u and v')
    out = []
    for (av, au), c in sorted(res.terms.items()):
        c = np.broadcast_to(np.asarray(c, dtype=float), geo.G)
        if np.any(c != 0.0):
            out.append((tuple(reversed(av)), tuple(reversed(au)), np.ascontiguousarray(c)))
    if not out:
        raise ValueError('the form has no non-zero coefficient')
    return out


# ---------------------------------------------------------------------------------------------
# passes: the device holds TWO derivative orders per axis (slots 0 and 1 of the basis table)
SLOT_SETS = ((0, 1), (0, 2), (1, 2))
MAX_TERMS_PER_CALL = 16         # igx_patch_set_pform
MAX_TERMS_PER_LAST_TYPE = 9     # the sum-factorised stage B takes at most nine terms per (test, trial) slot pair of the last axis


def plan_passes(terms, d):
    """Group the terms into device passes.  Returns a list of ``(slot0, slot1, [(mask_v, mask_u, c), ...])``: the derivative
    orders the two slots of every axis hold during the pass and its terms with their slot masks over the grid axes (bit a: the
    function takes slot 1 on grid axis a); terms of a pass with the same masks are summed; a pass has at most
    MAX_TERMS_PER_CALL terms.  The slot choices are picked greedily: the choice that covers most of the remaining terms
    first."""
    def fits(t, choice):
        return all(t[0][a] in choice[a] and t[1][a] in choice[a] for a in range(d))

    remaining = list(range(len(terms)))
    passes = []
    choices = list(itertools.product(SLOT_SETS, repeat=d))
    while remaining:
        best = max(choices, key=lambda ch: sum(fits(terms[i], ch) for i in remaining))
        took = [i for i in remaining if fits(terms[i], best)]
        assert took, 'a term fits no choice of slots'
        remaining = [i for i in remaining if i not in took]
        merged = {}
        for i in took:
            ov, ou, c = terms[i]
            mv = sum((best[a].index(ov[a])) << a for a in range(d))
            mu = sum((best[a].index(ou[a])) << a for a in range(d))
            _acc(merged, (mv, mu), c)
        items = sorted(merged.items())
        chunk, per_type = [], {}
        for (mv, mu), c in items:
            ty = ((mu >> (d - 1)) & 1) + 2 * ((mv >> (d - 1)) & 1)
            if len(chunk) == MAX_TERMS_PER_CALL or per_type.get(ty, 0) == MAX_TERMS_PER_LAST_TYPE:
                passes.append((tuple(s[0] for s in best), tuple(s[1] for s in best), chunk))
                chunk, per_type = [], {}
            chunk.append((mv, mu, c))
            per_type[ty] = per_type.get(ty, 0) + 1
        passes.append((tuple(s[0] for s in best), tuple(s[1] for s in best), chunk))
    return passes
