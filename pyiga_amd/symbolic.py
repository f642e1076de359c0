"""Tracing of coefficient arithmetic into C expressions, for forms whose coefficients are compiled for the device at run time
(``igx_patch_set_form_expr``; the reference generates and compiles code for every form: pyiga/codegen/cython.py:325-387,
pyiga/compile.py:58-73).

``pyiga_amd.forms`` evaluates a form string with numpy arrays on the Gauss grid.  Evaluated on a grid of ONE point whose
coordinates are the symbols below, the same code yields, instead of numbers, the expression tree of every coefficient field:
numpy's object arrays apply ``+ - * /``, ``einsum``, ``stack``, ``sum`` ... element by element, and ``np.sin(a)`` calls
``a.sin()``.  Python callables given as inputs are traced the same way (called with the symbols); whatever cannot be traced --
comparisons, ``np.where``, conversion to ``float``, integer tricks -- raises, and the caller falls back to sampling on the
host.  Constants fold in Python (IEEE doubles, the operations the device would do), ``0 + e`` and ``1 * e`` are dropped, and a
product with an exact zero is zero (the pruning of absent coefficients relies on it).
"""
import math

import numpy as np


def any_nonzero(a):
    """``np.any(a != 0.0)`` for the coefficient arrays of the form front-end: float arrays as usual; in an array of traced
    expressions only a constant 0 counts as zero (a non-constant expression is a present coefficient)."""
    a = np.asarray(a)
    if a.dtype != object:
        return bool(np.any(a != 0.0))
    for e in a.flat:
        if isinstance(e, Sym):
            if e.const is None or e.const != 0.0:
                return True
        elif e != 0.0:
            return True
    return False


class NotTraceable(TypeError):
    """The computation does something the tracer cannot express as a C expression in x, y, z."""


def _literal(v):
    v = float(v)
    if math.isnan(v) or math.isinf(v):
        raise NotTraceable('non-finite constant in a traced coefficient')
    return '(%r)' % v if v < 0 or (v == 0 and math.copysign(1.0, v) < 0) else repr(v)


class Sym:
    """A scalar double expression: C source text, or a constant."""
    __slots__ = ('c', 'const')
    __array_priority__ = 2000
    __array_ufunc__ = None

    def __init__(self, c, const=None):
        self.c, self.const = c, const

    @staticmethod
    def lift(v):
        if isinstance(v, Sym):
            return v
        if isinstance(v, (bool, np.bool_)) or not isinstance(v, (int, float, np.integer, np.floating)):
            raise NotTraceable('operand of type %s in a traced coefficient' % type(v).__name__)
        return Sym(_literal(v), float(v))

    # ---- arithmetic
    def _bin(self, o, op, swap=False):
        a, b = Sym.lift(self), Sym.lift(o)
        if swap:
            a, b = b, a
        if a.const is not None and b.const is not None:
            if op == '/' and b.const == 0.0:
                raise NotTraceable('division by zero in a traced coefficient')
            v = {'+': a.const + b.const, '-': a.const - b.const, '*': a.const * b.const, '/': a.const / b.const if op == '/' else 0.0}[op]
            return Sym(_literal(v), v)
        if op == '+':
            if a.const == 0.0: return b
            if b.const == 0.0: return a
        elif op == '-':
            if b.const == 0.0: return a
            if a.const == 0.0: return -b
        elif op == '*':
            if a.const == 0.0 or b.const == 0.0: return Sym('0.0', 0.0)
            if a.const == 1.0: return b
            if b.const == 1.0: return a
        elif op == '/':
            if b.const == 0.0:
                raise NotTraceable('division by zero in a traced coefficient')
            if a.const == 0.0: return Sym('0.0', 0.0)
            if b.const == 1.0: return a
        return Sym('(%s %s %s)' % (a.c, op, b.c))

    def __add__(self, o): return self._bin(o, '+')
    def __radd__(self, o): return self._bin(o, '+', True)
    def __sub__(self, o): return self._bin(o, '-')
    def __rsub__(self, o): return self._bin(o, '-', True)
    def __mul__(self, o): return self._bin(o, '*')
    def __rmul__(self, o): return self._bin(o, '*', True)
    def __truediv__(self, o): return self._bin(o, '/')
    def __rtruediv__(self, o): return self._bin(o, '/', True)
    def __pos__(self): return self

    def __neg__(self):
        if self.const is not None:
            return Sym(_literal(-self.const), -self.const)
        return Sym('(-%s)' % self.c)

    def __pow__(self, k):
        if isinstance(k, (int, np.integer)) and not isinstance(k, (bool, np.bool_)) and 0 <= int(k) <= 4:
            k = int(k)
            if k == 0:
                return Sym('1.0', 1.0)
            out = self
            for _ in range(k - 1):
                out = out * self                  # small integer powers as products
            return out
        e = Sym.lift(k)
        if self.const is not None and e.const is not None:
            return Sym.lift(self.const ** e.const)
        return Sym('pow(%s, %s)' % (self.c, e.c))

    def __rpow__(self, b):
        b = Sym.lift(b)
        return Sym('pow(%s, %s)' % (b.c, self.c))

    # ---- numpy ufuncs on object arrays call these
    def _fn(self, cname, pyf):
        if self.const is not None:
            return Sym.lift(pyf(self.const))
        return Sym('%s(%s)' % (cname, self.c))

    def sqrt(self): return self._fn('sqrt', math.sqrt)
    def exp(self): return self._fn('exp', math.exp)
    def log(self): return self._fn('log', math.log)
    def sin(self): return self._fn('sin', math.sin)
    def cos(self): return self._fn('cos', math.cos)
    def tan(self): return self._fn('tan', math.tan)
    def tanh(self): return self._fn('tanh', math.tanh)
    def sinh(self): return self._fn('sinh', math.sinh)
    def cosh(self): return self._fn('cosh', math.cosh)
    def arctan(self): return self._fn('atan', math.atan)
    def arcsin(self): return self._fn('asin', math.asin)
    def arccos(self): return self._fn('acos', math.acos)
    def log1p(self): return self._fn('log1p', math.log1p)
    def expm1(self): return self._fn('expm1', math.expm1)
    def log2(self): return self._fn('log2', math.log2)
    def log10(self): return self._fn('log10', math.log10)
    def cbrt(self): return self._fn('cbrt', lambda v: math.copysign(abs(v) ** (1.0 / 3.0), v))

    def _fn2(self, o, cname, pyf, swap=False):
        a, b = Sym.lift(self), Sym.lift(o)
        if swap:
            a, b = b, a
        if a.const is not None and b.const is not None:
            return Sym.lift(pyf(a.const, b.const))
        return Sym('%s(%s, %s)' % (cname, a.c, b.c))

    def arctan2(self, o): return self._fn2(o, 'atan2', math.atan2)
    def hypot(self, o): return self._fn2(o, 'hypot', math.hypot)
    def fabs(self): return self._fn('fabs', math.fabs)
    absolute = fabs
    __abs__ = fabs

    # ---- what the form front-end asks of a coefficient: "is it absent", "is it the same everywhere"
    # (only "is this coefficient identically zero" and comparisons of constants are answered; a comparison of an expression
    # with anything else is a branch on values the trace does not have: NotTraceable, the caller samples on the host)
    def __eq__(self, o):
        if isinstance(o, Sym):
            if self is o:
                return True
            if self.const is not None and o.const is not None:
                return self.const == o.const
            raise NotTraceable('comparison of traced expressions')
        if isinstance(o, (int, float, np.integer, np.floating)) and not isinstance(o, (bool, np.bool_)):
            if self.const is not None:
                return self.const == float(o)
            # (a user's ``np.where(x == 0, ...)`` or ``(x == 0) * c`` is a branch on values the trace does not have: the caller
            # samples on the host.  The front-end's own "is this coefficient absent" test is any_nonzero() below.)
            raise NotTraceable('comparison of a traced expression with a number')
        return NotImplemented

    def __ne__(self, o):
        r = self.__eq__(o)
        return r if r is NotImplemented else not r

    __hash__ = None

    def _no(self, *a, **k):
        raise NotTraceable('comparison / conversion of a traced coefficient')
    __lt__ = __le__ = __gt__ = __ge__ = __float__ = __int__ = __bool__ = __index__ = _no

    def __repr__(self):
        return 'Sym(%s)' % self.c


def coordinates(d):
    """Physical coordinates of the one-point grid: object array of shape (1,)*d + (d,) holding x, y[, z]."""
    X = np.empty((1,) * d + (d,), dtype=object)
    for k, name in enumerate('xyz'[:d]):
        X[(0,) * d + (k,)] = Sym(name)
    return X


def c_source(e):
    """C text of a traced scalar (the single entry of an object array, a Sym or a number)."""
    if isinstance(e, np.ndarray):
        if e.size != 1:
            raise NotTraceable('a traced coefficient that is not a scalar field')
        e = e.reshape(-1)[0]
    return Sym.lift(e).c


def trace_function(f, d):
    """C text of the scalar function `f(x, y[, z])` (a plain Python callable), or None when it cannot be traced (or is not
    scalar-valued): the caller samples it on the host then."""
    import os
    if os.environ.get('IGX_FORM_RTC', '1') == '0' or not callable(f) or hasattr(f, 'grid_eval'):
        return None
    try:
        X = coordinates(d)
        val = f(*(X[..., k] for k in range(d)))
        if isinstance(val, (tuple, list)):
            return None
        return c_source(np.broadcast_to(np.asarray(val, dtype=object), (1,) * d))
    except Exception:
        return None
