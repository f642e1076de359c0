"""Pin the CPU oracle (oracle/) against the reference's fixtures and golden vectors.

CPU only.  Mirrors test/test_assemble.py:10-40,83-100,138-168, test/test_bspline.py,
test/test_geometry.py:108-139 and test/test_mlmatrix.py:9-30 of the reference.
"""
import os

import numpy as np
import pytest
import scipy.sparse

from conftest import GOLDEN, golden_csr, rel_maxdiff, form_tables, form2d_cases, form_inputs


def test_make_knots_bits(oracle, golden):
    g = golden('knots')
    for key in g.files:
        if not key.startswith('p') or key in ('p3_n5_m2', 'p2_ab'):
            continue
        p, n = (int(x[1:]) for x in key.split('_'))
        assert np.array_equal(oracle.make_knots(p, 0.0, 1.0, n).kv, g[key]), key
    assert np.array_equal(oracle.make_knots(3, 0.0, 1.0, 5, mult=2).kv, g['p3_n5_m2'])
    assert np.array_equal(oracle.make_knots(2, -1.5, 2.25, 9).kv, g['p2_ab'])


def _bspline_cases(g):
    return sorted({k[:-3] for k in g.files if k.endswith('_kv')})


def test_active_deriv_and_tables(oracle, golden):
    g = golden('bspline')
    for name in _bspline_cases(g):
        kv = oracle.KnotVector(g[name + '_kv'], int(g[name + '_p']))
        nodes, weights = oracle.gauss_rule(kv.p + 1, kv.mesh[:-1], kv.mesh[1:])
        assert np.array_equal(nodes, g[name + '_nodes'])
        assert np.array_equal(weights, g[name + '_weights'])
        assert np.array_equal(kv.mesh, g[name + '_mesh'])
        der = oracle.active_deriv(kv, nodes, 1)
        # same algorithm; the reference is built with -ffast-math (setup.py:12), so
        # agreement is to a few ulp of the largest value, not bitwise
        assert np.abs(der - g[name + '_deriv']).max() <= 4e-16 * np.abs(g[name + '_deriv']).max(), name
        spans = np.array([oracle.findspan(kv.kv, kv.p, u) for u in nodes])
        assert np.array_equal(spans, g[name + '_spans'])
        assert np.array_equal(kv.mesh_support_idx_all(), g[name + '_meshsupp'])
        C = oracle.compute_values_derivs(kv, nodes, 1)
        assert C.shape == g[name + '_C'].shape
        assert np.array_equal(C == 0.0, g[name + '_C'] == 0.0)      # same zero structure
        assert np.abs(C - g[name + '_C']).max() <= 4e-16 * np.abs(g[name + '_C']).max()
        assert np.array_equal(oracle.compute_sparsity_ij(kv, kv), g[name + '_sparsity_ij'])


def test_sparsity_order(oracle, golden):
    g = golden('sparsity')
    for name, d in (('d3_p2_n3', 3), ('d2_p3_n4', 2), ('d2_mixed', 2), ('d3_mult', 3)):
        kvs = [oracle.KnotVector(g['%s_kv%d' % (name, k)], int(g['%s_p%d' % (name, k)])) for k in range(d)]
        bidx = [oracle.compute_sparsity_ij(kv, kv) for kv in kvs]
        bs = [(kv.numdofs, kv.numdofs) for kv in kvs]
        for lt in (0, 1):
            I, J = oracle.ml_nonzero(bidx, bs, lower_tri=bool(lt))
            assert np.array_equal(I, g['%s_lt%d_I' % (name, lt)])
            assert np.array_equal(J, g['%s_lt%d_J' % (name, lt)])


GEOS = ['quarter_annulus', 'bspline_quarter_annulus', 'twisted_box', 'cylinder', 'unit_square', 'unit_cube']


def _golden_geo(oracle, g, name):
    d = g[name + '_coeffs'].ndim - 1
    kvs = [oracle.KnotVector(g['%s_gkv%d' % (name, k)], int(g['%s_gp%d' % (name, k)])) for k in range(d)]
    return dict(kvs=kvs, coeffs=g[name + '_coeffs'], nurbs=bool(g[name + '_nurbs']))


@pytest.mark.parametrize('name', GEOS)
def test_geometry_jacobian(oracle, golden, name):
    g = golden('geometry')
    geo = _golden_geo(oracle, g, name)
    d = len(geo['kvs'])
    grid = [g['%s_grid%d' % (name, k)] for k in range(d)]
    jac = oracle.grid_jacobian(geo, grid)
    assert jac.shape == g[name + '_jac'].shape
    assert np.abs(jac - g[name + '_jac']).max() < 1e-14
    # own constructors reproduce the reference control nets
    ctor = {'quarter_annulus': oracle.geo_quarter_annulus, 'bspline_quarter_annulus': oracle.geo_bspline_quarter_annulus,
            'twisted_box': oracle.geo_twisted_box, 'cylinder': oracle.geo_cylinder,
            'unit_square': lambda: oracle.geo_unit_cube(2), 'unit_cube': lambda: oracle.geo_unit_cube(3)}[name]()
    assert np.allclose(ctor['coeffs'], geo['coeffs'], rtol=0, atol=1e-15)
    for a, b in zip(ctor['kvs'], geo['kvs']):
        assert a.p == b.p and np.array_equal(a.kv, b.kv)


def test_geometry_literal_jacobians(oracle):
    # literal values of test/test_geometry.py:108-121 (bspline_quarter_annulus)
    geo = oracle.geo_bspline_quarter_annulus()
    x = np.array([0.0, 0.5, 1.0])
    jac = oracle.grid_jacobian(geo, (x, x))
    assert np.allclose(jac[0, 0], [[1.0, 0.0], [0.0, 2.0]], atol=1e-14)
    assert np.allclose(jac[2, 2], [[0.0, -4.0], [1.0, 0.0]], atol=1e-14)


FIXTURES = [('d2_p3_n15', 'mass', 2), ('d2_p3_n15', 'stiff', 2), ('d3_p2_n10', 'mass', 3), ('d3_p2_n10', 'stiff', 3)]


@pytest.mark.parametrize('tag,kind,d', FIXTURES)
def test_reference_fixtures(oracle, tag, kind, d):
    """test/test_assemble.py:138-168 -- tolerance 1e-14 absolute, as in the reference."""
    if d == 2:
        kv = oracle.make_knots(3, 0.0, 1.0, 15)
        geo = oracle.geo_bspline_quarter_annulus()
    else:
        kv = oracle.make_knots(2, 0.0, 1.0, 10)
        geo = oracle.geo_twisted_box()
    A = oracle.assemble('mass' if kind == 'mass' else 'stiffness', (kv,) * d, geo)
    A_ref = oracle.read_sparse_matrix(os.path.join(GOLDEN, 'poisson_neu_%s_%s.mtx.gz' % (tag, kind)))
    assert abs(A - A_ref).max() < 1e-14
    assert abs(A - A.T).max() == 0.0


def test_golden_matrices(oracle, golden):
    g = golden('matrices')
    mk = oracle.make_knots
    cases = [
        ('d2_p3_n8_annulus_mass', 'mass', (mk(3, 0., 1., 8),) * 2, oracle.geo_quarter_annulus(), False),
        ('d2_p3_n8_annulus_stiff', 'stiffness', (mk(3, 0., 1., 8),) * 2, oracle.geo_quarter_annulus(), False),
        ('d2_p43_n56_square_mass', 'mass', (mk(4, 0., 1., 5), mk(3, 0., 1., 6)), oracle.geo_unit_cube(2), False),
        ('d2_p43_n56_square_stiff', 'stiffness', (mk(4, 0., 1., 5), mk(3, 0., 1., 6)), oracle.geo_unit_cube(2), False),
        ('d2_mult_annulus_mass', 'mass', (mk(3, 0., 1., 5, mult=2), mk(2, 0., 1., 4)), oracle.geo_bspline_quarter_annulus(), False),
        ('d2_mult_annulus_stiff', 'stiffness', (mk(3, 0., 1., 5, mult=2), mk(2, 0., 1., 4)), oracle.geo_bspline_quarter_annulus(), False),
        ('d3_p4_n3_cyl_stiff_lower', 'stiffness', (mk(4, 0., 1., 3),) * 3, oracle.geo_cylinder(), True),
        ('d3_p5_n3_cyl_stiff_lower', 'stiffness', (mk(5, 0., 1., 3),) * 3, oracle.geo_cylinder(), True),
        ('d3_p323_n342_tbox_mass_lower', 'mass', (mk(3, 0., 1., 3), mk(2, 0., 1., 4), mk(3, 0., 1., 2)), oracle.geo_twisted_box(), True),
        ('d3_p323_n342_tbox_stiff_lower', 'stiffness', (mk(3, 0., 1., 3), mk(2, 0., 1., 4), mk(3, 0., 1., 2)), oracle.geo_twisted_box(), True),
        ('d3_mult_cyl_stiff_lower', 'stiffness', (mk(2, 0., 1., 3, mult=2), mk(2, 0., 1., 3), mk(3, 0., 1., 2)), oracle.geo_cylinder(), True),
        ('d3_mult_cyl_mass_lower', 'mass', (mk(2, 0., 1., 3, mult=2), mk(2, 0., 1., 3), mk(3, 0., 1., 2)), oracle.geo_cylinder(), True),
    ]
    # round 6: repeated knots on the mid / the last axis, one or two degrees below nqp on the mid and last axes
    r6 = {'d3_midmult_cyl': (mk(3, 0., 1., 3), mk(3, 0., 1., 5, mult=2), mk(3, 0., 1., 4)),
          'd3_p443_cyl': (mk(4, 0., 1., 3), mk(4, 0., 1., 4), mk(3, 0., 1., 5)),
          'd3_p433_cyl': (mk(4, 0., 1., 3), mk(3, 0., 1., 5), mk(3, 0., 1., 5)),
          'd3_lastmult_cyl': (mk(3, 0., 1., 3), mk(3, 0., 1., 4), mk(3, 0., 1., 5, mult=2)),
          'd3_p424_cyl': (mk(4, 0., 1., 3), mk(2, 0., 1., 5), mk(4, 0., 1., 4))}
    for name, kvs in r6.items():
        cases.append((name + '_stiff_lower', 'stiffness', kvs, oracle.geo_cylinder(), True))
        cases.append((name + '_mass_lower', 'mass', kvs, oracle.geo_cylinder(), True))
    for name, kind, kvs, geo, lower in cases:
        A = oracle.assemble(kind, kvs, geo)
        if lower:
            A = scipy.sparse.tril(A, format='csr')
        R = golden_csr(g, name)
        assert A.shape == R.shape
        assert rel_maxdiff(A, R) < 1e-14, name


def test_golden_entries(oracle, golden):
    g = golden('entries')
    kv = oracle.make_knots(2, 0., 1., 4)
    for name, kind in (('stiff3d', 'stiffness'), ('mass3d', 'mass')):
        asm = oracle.Assembler(kind, (kv,) * 3, oracle.geo_cylinder())
        idx = g[name + '_idx']
        ref = g[name + '_multi']
        assert np.array_equal(ref, g[name + '_single'])
        out = asm.multi_entries(idx)
        scale = np.abs(ref).max()
        assert np.abs(out - ref).max() <= 1e-14 * scale
        assert np.array_equal(out == 0.0, ref == 0.0)          # out-of-pattern pairs give exact 0.0
        assert np.abs(asm.entries_python(idx) - ref).max() <= 1e-14 * scale
        assert np.array_equal(asm.multi_entries(idx, nthreads=4), out)   # thread-count independent
        # the box-local form used at the BASELINE sizes (fields only on a pair's support intersection): same sums, same bits
        assert np.array_equal(oracle.local_entries(kind, (kv,) * 3, oracle.geo_cylinder(), idx), out)
    kv2 = oracle.make_knots(3, 0., 1., 6)
    for name, kind in (('stiff2d', 'stiffness'), ('mass2d', 'mass')):
        asm = oracle.Assembler(kind, (kv2,) * 2, oracle.geo_quarter_annulus())
        out = asm.multi_entries(g[name + '_idx'])
        ref = g[name + '_multi']
        assert np.abs(out - ref).max() <= 1e-14 * np.abs(ref).max()
        assert np.abs(asm.entries_python(g[name + '_idx']) - ref).max() <= 1e-14 * np.abs(ref).max()
        assert np.array_equal(oracle.local_entries(kind, (kv2,) * 2, oracle.geo_quarter_annulus(), g[name + '_idx']), out)


def test_1d_literal_and_kron(oracle, golden):
    g = golden('kron')
    kv = oracle.KnotVector(g['kv1d'], 4)
    M = oracle.bsp_mixed_deriv_biform_1d(kv, 0, 0).toarray()
    K = oracle.bsp_mixed_deriv_biform_1d(kv, 1, 1).toarray()
    assert np.abs(M - g['M1d']).max() < 1e-15
    assert np.abs(K - g['K1d']).max() < 1e-13
    # two literal entries from test/test_assemble.py:12-13,27-28
    assert abs(M[0, 0] - 2.77777778e-02) < 1e-10 and abs(K[0, 0] - 9.1428571429) < 1e-10
    mk = oracle.make_knots
    kvs2 = (mk(4, 0., 1., 10), mk(3, 0., 1., 12))
    kvs3 = (mk(3, 0., 1., 4), mk(3, 0., 1., 5), mk(3, 0., 1., 6))
    for name, kind, kvs in (('kron2d_stiff', 'stiffness', kvs2), ('kron2d_mass', 'mass', kvs2),
                            ('kron3d_stiff', 'stiffness', kvs3), ('kron3d_mass', 'mass', kvs3)):
        A = oracle.kron_assemble(kind, kvs)
        assert rel_maxdiff(A, golden_csr(g, name)) < 1e-14
    # geometry path == Kronecker path on the identity map (test/test_assemble.py:83-100)
    A = oracle.assemble('stiffness', kvs2, oracle.geo_unit_cube(2))
    assert np.allclose(A.toarray(), oracle.kron_assemble('stiffness', kvs2).toarray(), rtol=0, atol=1e-14)
    A = oracle.assemble('stiffness', kvs3, oracle.geo_unit_cube(3))
    assert np.allclose(A.toarray(), oracle.kron_assemble('stiffness', kvs3).toarray(), rtol=0, atol=1e-14)


def test_convdiff_custom_form(oracle, golden):
    """Row f1 of SURVEY section 8: the run-time compiled convection-diffusion form (non-symmetric)."""
    g = golden('convdiff')
    mk = oracle.make_knots
    cases = [('d3_p2_n3_cyl', (mk(2, 0., 1., 3),) * 3, oracle.geo_cylinder()),
             ('d3_p32_n243_tbox', (mk(3, 0., 1., 2), mk(2, 0., 1., 4), mk(2, 0., 1., 3)), oracle.geo_twisted_box()),
             ('d3_mult_cyl', (mk(2, 0., 1., 3, mult=2), mk(2, 0., 1., 3), mk(3, 0., 1., 2)), oracle.geo_cylinder())]
    for name, kvs, geo in cases:
        A = oracle.assemble_nonsymmetric('convdiff', kvs, geo, coeff=lambda x, y, z: 1.0 + x)
        R = golden_csr(g, name)
        assert A.shape == R.shape
        assert rel_maxdiff(A, R) < 1e-14, name
        assert abs(A - A.T).max() > 1e-3 * abs(A).max()           # genuinely non-symmetric
        asm = oracle.Assembler('convdiff', kvs, geo, coeff=lambda x, y, z: 1.0 + x)
        out = asm.multi_entries(g[name + '_idx'])
        ref = g[name + '_multi']
        assert np.abs(out - ref).max() <= 1e-14 * np.abs(ref).max()
        assert np.array_equal(out == 0.0, ref == 0.0)


def test_inner_products_oracle_vs_reference(oracle, golden):
    """Load vectors: the oracle's restatement of inner_products against vectors produced by
    the reference (pyiga/assemble.py:288-340; recipes of test/test_assemble.py:223-245)."""
    g = golden('rhs')

    def f3(x, y, z):
        return np.cos(x) * np.exp(y) * np.sin(z)

    def f2(x, y):
        return np.exp(x + y)
    kvs = [oracle.make_knots(p, 0.0, 1.0, 8 + p) for p in range(3, 6)]
    tbox, ann, cyl = oracle.geo_twisted_box(), oracle.geo_quarter_annulus(), oracle.geo_cylinder()
    kv3 = (oracle.make_knots(2, 0.0, 1.0, 5), oracle.make_knots(3, 0.0, 1.0, 4, mult=2), oracle.make_knots(2, 0.0, 1.0, 6))
    kv2 = (oracle.make_knots(3, 0.0, 1.0, 6), oracle.make_knots(2, 0.0, 1.0, 5))
    cases = [('d3_param', oracle.inner_products(kvs, f3)),
             ('d3_tbox', oracle.inner_products(kvs, f3, geo=tbox)),
             ('d3_tbox_phys', oracle.inner_products(kvs, f3, f_physical=True, geo=tbox)),
             ('d3_tbox_asm', oracle.inner_products(kvs, f3, geo=tbox)),
             ('d3_tbox_phys_asm', oracle.inner_products(kvs, f3, f_physical=True, geo=tbox)),
             ('d3_cyl_phys', oracle.inner_products(kv3, f3, f_physical=True, geo=cyl)),
             ('d2_param', oracle.inner_products(kv2, f2)),
             ('d2_ann', oracle.inner_products(kv2, f2, geo=ann)),
             ('d2_ann_phys', oracle.inner_products(kv2, f2, f_physical=True, geo=ann)),
             ('d2_ann_vec_phys', oracle.inner_products(kv2, lambda x, y: (x * y, x - y), f_physical=True, geo=ann)),
             ('d1_param', oracle.inner_products((oracle.make_knots(3, 0.0, 1.0, 7),), lambda x: 1 + x ** 2))]
    for name, r in cases:
        assert r.shape == g[name].shape, name
        assert np.abs(r - g[name]).max() <= 1e-14 * np.abs(g[name]).max(), name


def test_general_forms_oracle_vs_reference(oracle, golden):
    """General scalar forms in the jets of u and v: the oracle's entry-wise restatement with
    hand-written coefficient tables against the matrices the reference compiled from the form strings."""
    g = golden('forms')
    spaces = {
        'cyl_p2': ((oracle.make_knots(2, 0.0, 1.0, 3),) * 3, oracle.geo_cylinder()),
        'tbox_mixed': ((oracle.make_knots(3, 0.0, 1.0, 2), oracle.make_knots(2, 0.0, 1.0, 4, mult=2),
                        oracle.make_knots(1, 0.0, 1.0, 3)), oracle.geo_twisted_box()),
    }
    for sname, (kvs, geo) in spaces.items():
        for fname, table in form_tables().items():
            A = oracle.assemble_nonsymmetric('form', kvs, geo, table=table, nthreads=4)
            R = golden_csr(g, '%s_%s' % (sname, fname))
            assert A.nnz == R.nnz and np.array_equal(A.indices, R.indices)
            assert rel_maxdiff(A, R) <= 1e-14, (sname, fname, rel_maxdiff(A, R))


def test_general_forms_2d_oracle_vs_reference(oracle, golden):
    g = golden('forms')
    kvs = (oracle.make_knots(3, 0.0, 1.0, 4), oracle.make_knots(2, 0.0, 1.0, 5, mult=2))
    for fname, (_, _, table) in form2d_cases().items():
        full = [[None] * 4 for _ in range(4)]
        for r in range(3):
            for s in range(3):
                full[r][s] = table[r][s]
        A = oracle.assemble_nonsymmetric('form', kvs, oracle.geo_quarter_annulus(), table=full, nthreads=2)
        R = golden_csr(g, 'd2_%s' % fname)
        assert A.nnz == R.nnz and np.array_equal(A.indices, R.indices)
        assert rel_maxdiff(A, R) <= 1e-14, (fname, rel_maxdiff(A, R))


def test_functionals_with_gradients_oracle_vs_reference(oracle, golden):
    """Arity-1 forms with derivatives of v ('inner(b, grad(v)) * dx'): oracle restatement vs the reference."""
    g = golden('forms')
    b3 = form_inputs()['b']
    kvs = (oracle.make_knots(3, 0.0, 1.0, 2), oracle.make_knots(2, 0.0, 1.0, 4, mult=2), oracle.make_knots(1, 0.0, 1.0, 3))
    r = oracle.load_vector_jet(kvs, oracle.geo_twisted_box(), [None] + [(lambda x, y, z, i=i: b3(x, y, z)[i]) for i in range(3)])
    assert np.abs(r - g['funcgrad_d3']).max() <= 1e-14 * np.abs(g['funcgrad_d3']).max()
    b2 = form2d_cases()['full'][1]['b']
    kv2 = (oracle.make_knots(3, 0.0, 1.0, 6), oracle.make_knots(2, 0.0, 1.0, 5))
    r = oracle.load_vector_jet(kv2, oracle.geo_quarter_annulus(),
                               [lambda x, y: x * y ** 2, lambda x, y: b2(x, y)[0], lambda x, y: b2(x, y)[1]])
    assert np.abs(r - g['funcgrad_d2']).max() <= 1e-14 * np.abs(g['funcgrad_d2']).max()
