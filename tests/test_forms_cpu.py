"""CPU tests of the form-string front-end (pyiga_amd/forms.py): pure host logic, no device.

Every string of the golden set must evaluate to exactly the coefficient table written out by hand in
conftest.py (the tables the oracle is pinned with), and unsupported constructs must be refused."""
import numpy as np
import pytest

from conftest import FORMS, form_inputs, form_tables, form2d_cases


@pytest.fixture(scope='module')
def forms():
    from pyiga_amd import forms
    return forms


def _eval_table(table, X):
    d = X.shape[-1]
    G = X.shape[:-1]
    out = [[None] * (d + 1) for _ in range(d + 1)]
    for r in range(d + 1):
        for s in range(d + 1):
            e = table[r][s]
            if e is not None:
                out[r][s] = np.broadcast_to(e(*(X[..., k] for k in range(d))) if callable(e) else e, G)
    return out


def _same(T, H):
    for r in range(len(H)):
        for s in range(len(H)):
            if H[r][s] is None or not np.any(H[r][s]):
                assert T[r][s] is None, (r, s)
            else:
                assert T[r][s] is not None, (r, s)
                assert np.abs(T[r][s] - H[r][s]).max() <= 1e-15 * max(1.0, np.abs(H[r][s]).max()), (r, s)


def test_strings_give_the_hand_written_tables_3d(forms):
    G = (4, 3, 5)
    X = np.random.default_rng(0).random(G + (3,)) + 0.5
    inp = form_inputs()
    tables = form_tables()
    for name, (expr, names) in FORMS.items():
        T = forms.coefficient_table(expr, G, X, {k: inp[k] for k in names})
        _same(T, _eval_table(tables[name], X))
        assert forms.arity(expr) == 2


def test_strings_give_the_hand_written_tables_2d(forms):
    G = (6, 4)
    X = np.random.default_rng(1).random(G + (2,)) + 0.5
    for name, (expr, inputs, table) in form2d_cases().items():
        _same(forms.coefficient_table(expr, G, X, inputs), _eval_table(table, X))


def test_builtin_strings(forms):
    G = (3, 3, 3)
    X = np.random.default_rng(2).random(G + (3,))
    T = forms.coefficient_table('inner(grad(u), grad(v)) * dx', G, X, {})
    assert all((T[r][s] is not None) == (r == s and r > 0) for r in range(4) for s in range(4))
    assert all(np.all(T[k][k] == 1.0) for k in (1, 2, 3))
    T = forms.coefficient_table('u * v * dx', G, X, {})
    assert T[0][0] is not None and np.all(T[0][0] == 1.0) and sum(e is not None for row in T for e in row) == 1
    # the form of BASELINE config 5
    cd = '(inner(diff_coeff*grad(u),grad(v)) + inner((x[1],-x[0],1.0),grad(u))*v)*dx'
    T = forms.coefficient_table(cd, G, X, dict(diff_coeff=lambda x, y, z: 1.0 + x))
    assert np.array_equal(T[0][1], X[..., 1]) and np.array_equal(T[0][2], -X[..., 0]) and np.all(T[0][3] == 1.0)
    assert all(np.array_equal(T[k][k], 1.0 + X[..., 0]) for k in (1, 2, 3)) and T[0][0] is None and T[1][0] is None
    # constants, parameters, algebra
    T = forms.coefficient_table('(a * u * v - u * v / 4 + 2 * (u * v)) * dx', G, X, dict(a=3.0))
    assert np.allclose(T[0][0], 3.0 - 0.25 + 2.0)
    T = forms.coefficient_table('inner(dot(K, grad(u)), grad(v)) * dx', G, X, dict(K=np.diag([1.0, 2.0, 3.0])))
    assert all(np.all(T[k][k] == k) for k in (1, 2, 3)) and T[1][2] is None


def test_functionals(forms):
    G = (5, 4)
    X = np.random.default_rng(3).random(G + (2,))
    f = lambda x, y: x * y ** 2
    assert np.array_equal(forms.functional_coefficient('f * v * dx', G, X, dict(f=f)), f(X[..., 0], X[..., 1]))
    F = forms.functional_coefficient('(2 * f + x[0]) * v * dx', G, X, dict(f=f))
    assert np.allclose(F, 2 * f(X[..., 0], X[..., 1]) + X[..., 0], rtol=0, atol=1e-15)
    assert forms.arity('f * v * dx') == 1
    with pytest.raises(NotImplementedError):
        forms.functional_coefficient('inner((1.0, 2.0), grad(v)) * dx', G, X, {})
    jet = forms.functional_jet('(f * v + inner((1.0, x[0]), grad(v))) * dx', G, X, dict(f=f))
    assert np.array_equal(jet[0], f(X[..., 0], X[..., 1])) and np.all(jet[1] == 1.0) and np.array_equal(jet[2], X[..., 0])
    jet = forms.functional_jet('inner((0.0, 2.0), grad(v)) * dx', G, X, {})
    assert jet[0] is None and jet[1] is None and np.all(jet[2] == 2.0)


@pytest.mark.parametrize('bad', ['u * v', 'inner(grad(u), grad(u)) * dx', 'inner(grad(u), grad(v)) * u * dx',
                                 'u * dx(v) * dx', 'grad(u) * grad(v) * dx', 'inner(grad(u), v) * dx',
                                 'grad(c * u) * dx'])
def test_unsupported_forms_are_refused(forms, bad):
    G = (3, 3, 3)
    X = np.random.default_rng(4).random(G + (3,))
    with pytest.raises(NotImplementedError):
        forms.arity(bad)
        forms.coefficient_table(bad, G, X, dict(c=lambda x, y, z: x))


def test_unknown_names(forms):
    G = (2, 2, 2)
    X = np.zeros(G + (3,))
    with pytest.raises(ValueError):
        forms.coefficient_table('q * u * v * dx', G, X, {})
    with pytest.raises(ValueError):
        forms.arity('f * dx')


def test_tensor_forms_tables():
    """Front-end for vector-valued basis functions and boundary integrals (pyiga_amd/tforms.py): the coefficient tables of a
    few strings written out by hand.  table[p][q][r][s]: test component p with jet index r (0 value, 1.. = d/dx, d/dy, d/dz)
    against trial component q with jet index s."""
    from pyiga_amd import tforms
    G = (3, 4)
    X = np.random.default_rng(0).random(G + (2,))
    ar, me, tab, ncs = tforms.evaluate('inner(as_matrix([[2,1],[0,0]]).dot(u), v) * dx', G, X, {}, bfuns=[('u', 2), ('v', 2)])
    assert (ar, me, ncs) == (2, 'dx', (2, 2))
    assert np.all(tab[0][0][0][0] == 2.0) and np.all(tab[0][1][0][0] == 1.0)
    assert all(e is None for p, q in ((1, 0), (1, 1)) for row in tab[p][q] for e in row)
    ar, me, tab, ncs = tforms.evaluate('(inner(grad(u), grad(v)) + div(u) * div(v)) * dx', G, X, {}, bfuns=[('u', 2), ('v', 2)])
    assert np.all(tab[0][0][1][1] == 2.0) and np.all(tab[0][0][2][2] == 1.0) and np.all(tab[0][1][1][2] == 1.0)
    assert np.all(tab[1][0][2][1] == 1.0) and tab[0][1][2][1] is None and np.all(tab[1][1][2][2] == 2.0)
    f = lambda x, y: x * y ** 2
    ar, me, tab, ncs = tforms.evaluate('f * div(v) * dx', G, X, {'f': f}, bfuns=[('v', 2)])
    assert (ar, ncs) == (1, (2,)) and np.array_equal(tab[0][1], f(X[..., 0], X[..., 1])) and np.array_equal(tab[1][2], tab[0][1])
    assert tab[0][0] is None and tab[0][2] is None and tab[1][1] is None
    G3 = (2, 2)
    X3 = np.random.default_rng(1).random(G3 + (3,))
    n = np.zeros(G3 + (3,))
    n[..., 2] = 1.0
    ar, me, tab, ncs = tforms.evaluate('inner(cross(n, grad(u)), cross(n, grad(v))) * ds', G3, X3, {}, normal=n)
    t = tab[0][0]
    assert me == 'ds' and np.all(t[1][1] == 1.0) and np.all(t[2][2] == 1.0) and t[3][3] is None and t[0][0] is None
    ar, me, tab, ncs = tforms.evaluate('inner(v, n) * ds', G3, X3, {}, bfuns=[('v', 3)], normal=n)
    assert ar == 1 and tab[0][0] is None and tab[1][0] is None and np.all(tab[2][0] == 1.0)
    # the scalar front-end and this one agree on the convection-diffusion form of BASELINE config 5
    from pyiga_amd import forms
    G4 = (2, 3, 2)
    X4 = np.random.default_rng(2).random(G4 + (3,))
    s = '(inner(diff_coeff*grad(u),grad(v)) + inner((x[1],-x[0],1.0),grad(u))*v)*dx'
    dc = lambda x, y, z: 1.0 + x
    a = forms.coefficient_table(s, G4, X4, {'diff_coeff': dc})
    ar, me, tab, ncs = tforms.evaluate(s, G4, X4, {'diff_coeff': dc})
    for r in range(4):
        for c in range(4):
            assert (a[r][c] is None) == (tab[0][0][r][c] is None)
            if a[r][c] is not None:
                assert np.array_equal(a[r][c], tab[0][0][r][c])
    for bad in ('inner(grad(u), grad(u)) * dx', 'grad(c * u) * dx', 'u * v', 'inner(u, grad(v)) * dx'):
        with pytest.raises(NotImplementedError):
            tforms.evaluate(bad, G, X, {'c': lambda x, y: x}, bfuns=[('u', 2), ('v', 2)])


# ---------------------------------------------------------------------------------------------
# second derivatives / parametric derivatives (pyiga_amd/pforms.py): host logic only
def _polar_geometry(G):
    """x = (1 + s) cos t, y = (1 + s) sin t on a grid: coordinates, Jacobians and Hessians of the map, (x, y) index order."""
    t, s = np.meshgrid(np.linspace(0.1, 1.2, G[0]), np.linspace(0.0, 1.0, G[1]), indexing='ij')   # grid axes (y-like, x-like)
    r = 1.0 + s
    X = np.stack([r * np.cos(t), r * np.sin(t)], -1)
    Jac = np.empty(G + (2, 2))                       # Jac[i, k] = d x_i / d xi_k, xi = (s, t)
    Jac[..., 0, 0], Jac[..., 0, 1] = np.cos(t), -r * np.sin(t)
    Jac[..., 1, 0], Jac[..., 1, 1] = np.sin(t), r * np.cos(t)
    H2 = np.zeros(G + (2, 2, 2))                     # H2[m, e, u]
    H2[..., 0, 0, 1] = H2[..., 0, 1, 0] = -np.sin(t)
    H2[..., 0, 1, 1] = -r * np.cos(t)
    H2[..., 1, 0, 1] = H2[..., 1, 1, 0] = np.cos(t)
    H2[..., 1, 1, 1] = -r * np.sin(t)
    return X, Jac, H2


def test_physical_second_derivatives_transform():
    """The parametric terms of a physical second derivative reproduce the Hessian of f(x, y) = x^2 y + 3 x y^2 from the
    parametric derivatives of f o F (chain rule forwards, transformation backwards)."""
    from pyiga_amd import pforms
    G = (5, 4)
    X, Jac, H2 = _polar_geometry(G)
    x, y = X[..., 0], X[..., 1]
    f1 = np.stack([2 * x * y + 3 * y ** 2, x ** 2 + 6 * x * y], -1)
    f2 = np.empty(G + (2, 2))
    f2[..., 0, 0], f2[..., 0, 1], f2[..., 1, 0], f2[..., 1, 1] = 2 * y, 2 * x + 6 * y, 2 * x + 6 * y, 6 * x
    gp = np.einsum('...i,...ik->...k', f1, Jac)
    Hp = np.einsum('...ij,...ie,...ju->...eu', f2, Jac, Jac) + np.einsum('...i,...ieu->...eu', f1, H2)
    geo = pforms._Geo(G, Jac, H2)
    for (i, j) in ((0, 0), (0, 1), (1, 1)):
        jet = pforms._Jet('u', {((0, 0), False): 1.0}).derivative(i, 1, False).derivative(j, 1, False)
        val = 0.0
        for alpha, c in jet.parametric_terms(geo).items():
            ks = [k for k in range(2) for _ in range(alpha[k])]
            val = val + c * (gp[..., ks[0]] if len(ks) == 1 else Hp[..., ks[0], ks[1]])
        assert np.abs(val - f2[..., i, j]).max() <= 1e-12 * np.abs(f2).max()
    # first derivatives likewise
    for i in range(2):
        jet = pforms._Jet('u', {((0, 0), False): 1.0}).derivative(i, 1, False)
        val = sum(c * gp[..., [k for k in range(2) if alpha[k]][0]] for alpha, c in jet.parametric_terms(geo).items())
        assert np.abs(val - f1[..., i]).max() <= 1e-12 * np.abs(f1).max()


def test_pform_terms_and_passes():
    from pyiga_amd import pforms
    G = (3, 4)
    X = np.stack(np.meshgrid(np.linspace(0, 1, 3), np.linspace(0, 1, 4), indexing='ij')[::-1], -1)
    Jac = np.broadcast_to(np.eye(2), G + (2, 2)).copy()
    H2 = np.zeros(G + (2, 2, 2))
    # identity geometry: physical = parametric; orders are per GRID axis (x last)
    t = pforms.evaluate('inner(hess(u), hess(v)) * dx', G, X, Jac, H2, {})
    assert sorted((ov, ou, float(c.flat[0])) for ov, ou, c in t) == [((0, 2), (0, 2), 1.0), ((1, 1), (1, 1), 2.0), ((2, 0), (2, 0), 1.0)]
    t = pforms.evaluate('(Dx(u, 0, times=2) * v + 2 * u * Dx(v, 1, parametric=True)) * dx', G, X, 2.0 * Jac, H2, {})
    got = {(ov, ou): float(c.flat[0]) for ov, ou, c in t}
    assert got == {((0, 0), (0, 2)): 1.0, ((1, 0), (0, 0)): 8.0}          # |det| = 4, d/dx = (1/2) d/dxi
    t3 = pforms.evaluate('(c * tr(hess(u)) * v + inner(hess(u), hess(v)) + inner(grad(u), grad(v)) + u * v) * dx', (2, 2, 2),
                         np.zeros((2, 2, 2, 3)), np.broadcast_to(np.eye(3), (2, 2, 2, 3, 3)).copy(), np.zeros((2, 2, 2, 3, 3, 3)),
                         dict(c=lambda x, y, z: 2.0 + x))
    for terms, d in ((t, 2), (t3, 3)):
        passes = pforms.plan_passes(terms, d)
        seen = []
        for slot0, slot1, chunk in passes:
            assert 1 <= len(chunk) <= pforms.MAX_TERMS_PER_CALL and all(a < b for a, b in zip(slot0, slot1))
            per_type = {}
            for mv, mu, c in chunk:
                ov = tuple((slot1 if (mv >> a) & 1 else slot0)[a] for a in range(d))
                ou = tuple((slot1 if (mu >> a) & 1 else slot0)[a] for a in range(d))
                seen.append((ov, ou))
                ty = ((mu >> (d - 1)) & 1) + 2 * ((mv >> (d - 1)) & 1)
                per_type[ty] = per_type.get(ty, 0) + 1
            assert max(per_type.values()) <= pforms.MAX_TERMS_PER_LAST_TYPE
        assert sorted(seen) == sorted((ov, ou) for ov, ou, _ in terms)      # every term in exactly one pass
    for bad in ('Dx(u, 0, times=3) * v * dx', 'Dx(c * u, 0) * v * dx', 'Dx(Dx(u, 0, parametric=True), 1) * v * dx',
                'hess(u) * v * dx', 'hess(u)[0, 0] * hess(u)[1, 1] * dx', 'u * v'):
        with pytest.raises(NotImplementedError):
            pforms.evaluate(bad, G, X, Jac, H2, dict(c=lambda x, y: x))


# ---------------------------------------------------------------------------------------------
# forms compiled at run time: the tracer (pyiga_amd/symbolic.py) and the generated kernel (hiprtc cross-compiles without a GPU)
def _eval_c(src, X):
    ns = {k: getattr(np, k) for k in ('sin', 'cos', 'tan', 'exp', 'log', 'sqrt', 'tanh', 'sinh', 'cosh', 'fabs')}
    ns.update(atan=np.arctan, atan2=np.arctan2, hypot=np.hypot, asin=np.arcsin, log1p=np.log1p, pow=np.power, x=X[..., 0], y=X[..., 1], z=X[..., 2] if X.shape[-1] > 2 else 0.0)
    return eval(src, {'__builtins__': {}}, ns) + 0.0 * X[..., 0]


def test_traced_tables_equal_sampled_tables(forms):
    """Every form string of the golden set, traced into C expressions and evaluated again with numpy, gives the table the
    sampling front-end gives (same entries present, same values); inputs that cannot be traced are reported as such."""
    from pyiga_amd import symbolic
    G = (3, 4, 5)
    X = np.random.default_rng(8).random(G + (3,)) + 0.5
    inp = form_inputs()
    inp['w'] = lambda x, y, z: np.exp(-((x - 0.5) ** 2 + y ** 3) / 0.7) * np.sqrt(1.0 + z * z) + np.cos(x * y) / (2.0 + np.sin(z)) + abs(x - y) + np.arctan2(y, x) * np.hypot(x, z) + np.log1p(x) * np.arcsin(0.3 * y)
    cases = dict(FORMS)
    cases['transcendental'] = ('(w * inner(grad(u), grad(v)) + w**2 * u * v - inner((w, 0.0, 2.0), grad(u)) * v) * dx', ('w',))
    for fname, (form, names) in cases.items():
        kw = {k: inp[k] for k in names}
        sampled = forms.coefficient_table(form, G, X, kw)
        traced = forms.symbolic_table(form, 3, kw)
        for r in range(4):
            for s in range(4):
                assert (sampled[r][s] is None) == (traced[r][s] is None), (fname, r, s)
                if sampled[r][s] is not None:
                    assert np.allclose(_eval_c(traced[r][s], X), sampled[r][s], rtol=1e-15, atol=1e-15), (fname, r, s, traced[r][s])
    for bad in (lambda x, y, z: np.where(x > 0.5, 1.0, 2.0), lambda x, y, z: np.maximum(x, 0.3), lambda x, y, z: float(x.sum()),
                lambda x, y, z: np.where(x == 0.5, 1.0, 2.0), lambda x, y, z: np.where(x == y, 1.0, 2.0), lambda x, y, z: x // 2, lambda x, y, z: np.sign(x),
                lambda x, y, z: np.full(G, 2.0)):
        with pytest.raises(Exception) as e:
            forms.symbolic_table('c * u * v * dx', 3, dict(c=bad))
        assert not isinstance(e.value, NotImplementedError)
    with pytest.raises(NotImplementedError):                      # a variable coefficient inside grad(): not this front-end's
        forms.symbolic_table('inner(grad(c * u), grad(v)) * dx', 3, dict(c=lambda x, y, z: x))
    assert forms.symbolic_table('inner(grad(2 * u), grad(v)) * dx', 3, {})[1][1] == '2.0'
    # constants fold, zeros vanish
    S = symbolic.Sym
    x = S('x')
    assert (0.0 * x + 1.0 * x).c == 'x' and (x * 0.0).const == 0.0 and (S.lift(2.0) * 3.0 + 1.0).const == 7.0 and (-S.lift(2.0)).c == '(-2.0)'
    assert (x ** 2).c == '(x * x)' and (x ** 0.5).c == 'pow(x, 0.5)' and (x - 0.0) is x


def test_form_kernel_compiles_and_is_cached(forms, tmp_path, monkeypatch):
    """The generated kernel of a traced form: compiled by hiprtc into the cache (no GPU needed), found there the second time."""
    import ctypes
    from pyiga_amd import _lib
    cdll = _lib.load()
    monkeypatch.setenv('IGX_CACHE_DIR', str(tmp_path / 'cache'))
    inp = form_inputs()
    traced = forms.symbolic_table(FORMS['full'][0], 3, {k: inp[k] for k in FORMS['full'][1]})
    exprs = [e.encode() for row in traced for e in row if e is not None]
    arr = (ctypes.c_char_p * len(exprs))(*exprs)
    buf = ctypes.create_string_buffer(1024)
    hit = ctypes.c_int(-1)
    assert cdll.igx_rtc_compile_form(len(exprs), arr, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 0
    path = buf.value.decode()
    assert open(path, 'rb').read(4) == b'\x7fELF'
    assert cdll.igx_rtc_compile_form(len(exprs), arr, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 1
    assert cdll.igx_rtc_compile_form(len(exprs) - 1, arr, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 0 and buf.value.decode() != path


def test_field_kernel_of_a_traced_form_compiles(forms, tmp_path, monkeypatch):
    """The field kernel with the expressions inside (geometry + coefficients + jet transformation in one pass): generated for
    2D / 3D, B-spline / NURBS geometries, compiled by hiprtc without a GPU, cached; another geometry type is another kernel."""
    import ctypes
    from pyiga_amd import _lib
    cdll = _lib.load()
    monkeypatch.setenv('IGX_CACHE_DIR', str(tmp_path / 'cache'))
    inp = form_inputs()
    traced = forms.symbolic_table(FORMS['full'][0], 3, {k: inp[k] for k in FORMS['full'][1]})
    table = [e.encode() if e is not None else None for row in traced for e in row]
    assert len(table) == 16
    arr = (ctypes.c_char_p * 16)(*table)
    buf = ctypes.create_string_buffer(1024)
    hit = ctypes.c_int(-1)
    paths = set()
    for ncomp in (3, 4):
        assert cdll.igx_rtc_compile_form_fields(3, ncomp, arr, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0, _lib.last_error()
        assert hit.value == 0 and open(buf.value.decode(), 'rb').read(4) == b'\x7fELF'
        paths.add(buf.value.decode())
        assert cdll.igx_rtc_compile_form_fields(3, ncomp, arr, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 1
    # 2D: the coefficients of the third derivative direction do not exist
    t2 = [None] * 16
    t2[0] = b'x * y'; t2[5] = b'1.0 + x'; t2[6] = b'0.5'; t2[9] = b'0.5'; t2[10] = b'2.0 - y'; t2[1] = b'sin(pi * x)'
    arr2 = (ctypes.c_char_p * 16)(*t2)
    for ncomp in (2, 3):
        assert cdll.igx_rtc_compile_form_fields(2, ncomp, arr2, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0, _lib.last_error()
        paths.add(buf.value.decode())
    assert len(paths) == 4
    # a syntax error is reported with the compiler's message, bad arguments are refused
    t2[0] = b'x +* y'
    assert cdll.igx_rtc_compile_form_fields(2, 2, (ctypes.c_char_p * 16)(*t2), b'gfx950', buf, 1024, ctypes.byref(hit)) != 0
    assert 'error' in _lib.last_error()
    assert cdll.igx_rtc_compile_form_fields(2, 5, arr2, b'gfx950', buf, 1024, ctypes.byref(hit)) != 0
    assert cdll.igx_rtc_compile_form_fields(3, 3, (ctypes.c_char_p * 16)(*([None] * 16)), b'gfx950', buf, 1024, ctypes.byref(hit)) != 0


def test_load_vector_kernel_with_the_function_inside_compiles(tmp_path, monkeypatch):
    """The generated variant of the fused contraction kernel (function evaluated at the points of the grid line): every degree
    and line length class, physical and parametric coordinates, compiled by hiprtc without a GPU."""
    import ctypes
    from pyiga_amd import _lib, symbolic
    import numpy as np
    cdll = _lib.load()
    monkeypatch.setenv('IGX_CACHE_DIR', str(tmp_path / 'cache'))
    src = symbolic.trace_function(lambda x, y, z: np.cos(x) * np.exp(y) * np.sin(z) + x * y, 3).encode()
    buf = ctypes.create_string_buffer(1024)
    hit = ctypes.c_int(-1)
    paths = set()
    for P, npass, par in ((2, 2, 1), (3, 3, 0), (5, 3, 1), (5, 4, 0), (6, 2, 1)):
        assert cdll.igx_rtc_compile_load_vector(P, npass, par, src, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0, _lib.last_error()
        assert hit.value == 0 and open(buf.value.decode(), 'rb').read(4) == b'\x7fELF'
        paths.add(buf.value.decode())
    assert len(paths) == 5
    assert cdll.igx_rtc_compile_load_vector(5, 3, 1, src, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 1
    assert cdll.igx_rtc_compile_load_vector(7, 3, 1, src, b'gfx950', buf, 1024, ctypes.byref(hit)) != 0
    assert cdll.igx_rtc_compile_load_vector(5, 3, 1, b'x +', b'gfx950', buf, 1024, ctypes.byref(hit)) != 0


def test_equality_with_a_number_is_not_traced():
    """a callable that branches on ``x == 0`` must not be traced with the branch silently dropped."""
    from pyiga_amd import symbolic
    X = symbolic.coordinates(3)
    x = X[..., 0]
    with pytest.raises(symbolic.NotTraceable):
        np.where(x == 0, 1.0, np.sin(x) / x)
    with pytest.raises(symbolic.NotTraceable):
        (x != 0) * 2.0
    assert symbolic.trace_function(lambda x, y, z: np.where(x == 0, 1.0, x), 3) is None
    # the front-end's own test for an absent coefficient
    assert symbolic.any_nonzero(x) and not symbolic.any_nonzero(np.array([symbolic.Sym('0.0', 0.0)], dtype=object))
    assert symbolic.any_nonzero(np.array([0.0, 1e-300])) and not symbolic.any_nonzero(np.zeros(3))
