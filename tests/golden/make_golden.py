#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (c-f-h/pyiga).

Run in the build container only (the reference does not exist on the GPU box):

    # one-time scratch build of the unmodified reference, outside the repo
    mkdir -p /tmp/pyiga_oracle && cp -r /root/reference/{pyiga,setup.py,test} /tmp/pyiga_oracle
    (cd /tmp/pyiga_oracle && python3 setup.py build_ext -i)
    # then
    PYTHONPATH=/tmp/pyiga_oracle python3 tests/golden/make_golden.py

Writes `tests/golden/golden_*.npz` -- inputs and expected outputs only (data, no
reference source).  Every array is produced by calling the reference's public
API; the recipe for each case is recorded in the `desc` string stored with it.

Cases follow SURVEY.md section 8c.
"""
import os
import sys
import numpy as np
import scipy.sparse

import pyiga
from pyiga import bspline, geometry, assemble, assemblers, quadrature, mlmatrix
from pyiga import assemble_tools

pyiga.set_max_threads(1)
OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(OUT, 'golden_%s.npz' % name)
    np.savez_compressed(path, **arrays)
    print('wrote', path, {k: np.shape(v) for k, v in arrays.items()})


def cylinder():
    # quarter-annulus cylinder used by the 3D perf configs (test/test_assemble.py:325-327)
    return geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())


# ---------------------------------------------------------------------------
# (1) B-spline evaluation: active_deriv / findspan  (bspline_cy.pyx:13-145)
def golden_bspline():
    out = {}
    kvs = {}
    for p in (1, 2, 3, 4, 5):
        kvs['p%d_n4' % p] = bspline.make_knots(p, 0.0, 1.0, 4)
    kvs['p3_n5_mult2'] = bspline.make_knots(3, 0.0, 1.0, 5, mult=2)
    kvs['p4_custom'] = bspline.KnotVector(np.array(
        [0., 0., 0., 0., 0., 0.25, 0.35, 0.45, 0.55, 0.65, 0.9, 0.9, 0.9, 0.9, 0.9]), 4)
    kvs['p2_nonuniform'] = bspline.KnotVector(np.array(
        [0., 0., 0., 0.1, 0.15, 0.5, 0.5, 0.8, 1., 1., 1.]), 2)
    for name, kv in kvs.items():
        q = kv.p + 1
        nodes, weights = quadrature.make_iterated_quadrature(kv.mesh, q)
        der = np.asarray(bspline.active_deriv(kv, nodes, 1))   # (2, p+1, ng)
        spans = np.array([kv.findspan(u) for u in nodes])
        out[name + '_kv'] = kv.kv
        out[name + '_p'] = np.array(kv.p)
        out[name + '_nodes'] = nodes
        out[name + '_weights'] = weights
        out[name + '_deriv'] = der
        out[name + '_spans'] = spans
        out[name + '_meshsupp'] = kv.mesh_support_idx_all()
        out[name + '_mesh'] = kv.mesh
        # dense (ndofs, ngauss, 2) table the reference assemblers use
        out[name + '_C'] = assemble_tools.compute_values_derivs(kv, nodes, derivs=1)
        out[name + '_sparsity_ij'] = mlmatrix.compute_sparsity_ij(kv, kv)
    save('bspline', **out)


# ---------------------------------------------------------------------------
# (2) sparsity structure (mlmatrix.py:59-130, mlmatrix_cy.pyx:189-289)
def golden_sparsity():
    out = {}
    kv2 = bspline.make_knots(2, 0.0, 1.0, 3)
    kv3 = bspline.make_knots(3, 0.0, 1.0, 4)
    kvm = bspline.make_knots(2, 0.0, 1.0, 3, mult=2)
    for name, kvs in (('d3_p2_n3', (kv2, kv2, kv2)), ('d2_p3_n4', (kv3, kv3)),
                      ('d2_mixed', (kv2, kv3)), ('d3_mult', (kvm, kv2, kvm))):
        S = mlmatrix.MLStructure.from_kvs(kvs, kvs)
        for lt in (False, True):
            I, J = S.nonzero(lower_tri=lt)
            out['%s_lt%d_I' % (name, lt)] = np.asarray(I)
            out['%s_lt%d_J' % (name, lt)] = np.asarray(J)
        for k, kv in enumerate(kvs):
            out['%s_kv%d' % (name, k)] = kv.kv
            out['%s_p%d' % (name, k)] = np.array(kv.p)
    save('sparsity', **out)


# ---------------------------------------------------------------------------
# (3) geometry Jacobians on the Gauss grid (bspline.py:897-921, geometry.py:116-123)
def golden_geometry():
    out = {}
    geos = {
        'quarter_annulus': (geometry.quarter_annulus(), 3),
        'bspline_quarter_annulus': (geometry.bspline_quarter_annulus(), 3),
        'twisted_box': (geometry.twisted_box(), 2),
        'cylinder': (cylinder(), 4),
        'unit_square': (geometry.unit_square(), 2),
        'unit_cube': (geometry.unit_cube(), 2),
    }
    for name, (geo, p) in geos.items():
        kv = bspline.make_knots(p, 0.0, 1.0, 2)
        grid, w = quadrature.make_tensor_quadrature([kv.mesh] * geo.sdim, p + 1)
        out[name + '_coeffs'] = np.asarray(geo.coeffs)
        out[name + '_nurbs'] = np.array(isinstance(geo, geometry.NurbsFunc))
        for k, gkv in enumerate(geo.kvs):
            out[name + '_gkv%d' % k] = gkv.kv
            out[name + '_gp%d' % k] = np.array(gkv.p)
        for k in range(geo.sdim):
            out[name + '_grid%d' % k] = grid[k]
        out[name + '_jac'] = np.asarray(geo.grid_jacobian(grid))
        out[name + '_eval'] = np.asarray(geo.grid_eval(grid))
    save('geometry', **out)


# ---------------------------------------------------------------------------
# (4)+(5) assembler entries and full matrices
def lower_csr(A):
    L = scipy.sparse.tril(A.tocsr(), format='csr')
    L.sort_indices()
    return L


def put_matrix(out, name, A):
    A = A.tocsr()
    A.sort_indices()
    out[name + '_shape'] = np.array(A.shape)
    out[name + '_indptr'] = A.indptr
    out[name + '_indices'] = A.indices
    out[name + '_data'] = A.data


def golden_matrices():
    out = {}
    # -- 2D p=3 n=8 NURBS annulus, mass + stiffness (full)
    kv = bspline.make_knots(3, 0.0, 1.0, 8)
    geo = geometry.quarter_annulus()
    put_matrix(out, 'd2_p3_n8_annulus_mass', assemble.mass((kv, kv), geo))
    put_matrix(out, 'd2_p3_n8_annulus_stiff', assemble.stiffness((kv, kv), geo))
    # -- 2D unequal degrees p=(4,3) n=(5,6) unit square
    kvs = (bspline.make_knots(4, 0.0, 1.0, 5), bspline.make_knots(3, 0.0, 1.0, 6))
    put_matrix(out, 'd2_p43_n56_square_mass', assemble.mass(kvs, geometry.unit_square()))
    put_matrix(out, 'd2_p43_n56_square_stiff', assemble.stiffness(kvs, geometry.unit_square()))
    # -- 2D with interior knot multiplicity 2 on the B-spline annulus
    kvm = bspline.make_knots(3, 0.0, 1.0, 5, mult=2)
    kvn = bspline.make_knots(2, 0.0, 1.0, 4)
    geo_b = geometry.bspline_quarter_annulus()
    put_matrix(out, 'd2_mult_annulus_mass', assemble.mass((kvm, kvn), geo_b))
    put_matrix(out, 'd2_mult_annulus_stiff', assemble.stiffness((kvm, kvn), geo_b))
    out['d2_mult_kv0'] = kvm.kv
    out['d2_mult_kv1'] = kvn.kv
    # -- 3D p=4 n=3 and p=5 n=3 annulus cylinder stiffness (lower triangle only)
    cyl = cylinder()
    for p in (4, 5):
        kv = bspline.make_knots(p, 0.0, 1.0, 3)
        A = assemble.stiffness((kv, kv, kv), cyl)
        put_matrix(out, 'd3_p%d_n3_cyl_stiff_lower' % p, lower_csr(A))
    # -- 3D p=3 mixed n, twisted box: mass + stiffness (lower)
    kvs3 = (bspline.make_knots(3, 0.0, 1.0, 3), bspline.make_knots(2, 0.0, 1.0, 4),
            bspline.make_knots(3, 0.0, 1.0, 2))
    tb = geometry.twisted_box()
    put_matrix(out, 'd3_p323_n342_tbox_mass_lower', lower_csr(assemble.mass(kvs3, tb)))
    put_matrix(out, 'd3_p323_n342_tbox_stiff_lower', lower_csr(assemble.stiffness(kvs3, tb)))
    # -- 3D with multiplicity on one axis, cylinder
    kvs3m = (bspline.make_knots(2, 0.0, 1.0, 3, mult=2), bspline.make_knots(2, 0.0, 1.0, 3),
             bspline.make_knots(3, 0.0, 1.0, 2))
    put_matrix(out, 'd3_mult_cyl_stiff_lower', lower_csr(assemble.stiffness(kvs3m, cyl)))
    put_matrix(out, 'd3_mult_cyl_mass_lower', lower_csr(assemble.mass(kvs3m, cyl)))
    for k, kvx in enumerate(kvs3m):
        out['d3_mult_kv%d' % k] = kvx.kv
    # -- round 6: the shapes the fused second kernel (k_bf3) was generalised to, pinned to the reference itself: repeated knots on
    #    the MID axis (equal degrees), one or both of the mid / last axes one degree below nqp = max degree + 1
    #    (pyiga/assemblers.pyx:1338); and the two shapes round 6 moves onto it: repeated knots on the LAST axis, a degree gap of two
    mk = bspline.make_knots
    r6 = {'d3_midmult_cyl': (mk(3, 0.0, 1.0, 3), mk(3, 0.0, 1.0, 5, mult=2), mk(3, 0.0, 1.0, 4)),
          'd3_p443_cyl': (mk(4, 0.0, 1.0, 3), mk(4, 0.0, 1.0, 4), mk(3, 0.0, 1.0, 5)),
          'd3_p433_cyl': (mk(4, 0.0, 1.0, 3), mk(3, 0.0, 1.0, 5), mk(3, 0.0, 1.0, 5)),
          'd3_lastmult_cyl': (mk(3, 0.0, 1.0, 3), mk(3, 0.0, 1.0, 4), mk(3, 0.0, 1.0, 5, mult=2)),
          'd3_p424_cyl': (mk(4, 0.0, 1.0, 3), mk(2, 0.0, 1.0, 5), mk(4, 0.0, 1.0, 4))}
    for name, kvs in r6.items():
        put_matrix(out, name + '_stiff_lower', lower_csr(assemble.stiffness(kvs, cyl)))
        put_matrix(out, name + '_mass_lower', lower_csr(assemble.mass(kvs, cyl)))
    save('matrices', **out)

    # -- single entries via the assembler object (genericasm.pxi:677-758)
    out = {}
    kv = bspline.make_knots(2, 0.0, 1.0, 4)
    for name, asm in (('stiff3d', assemblers.StiffnessAssembler3D((kv, kv, kv), cyl)),
                      ('mass3d', assemblers.MassAssembler3D((kv, kv, kv), cyl))):
        N = kv.numdofs
        nd = N ** 3
        rng = np.random.default_rng(7)
        pairs = [(0, 0), (nd - 1, nd - 1), (nd // 2, nd // 2), (0, nd - 1), (nd - 1, 0),
                 (N * N + N + 1, 1), (5, 5 + N * N), (5 + N * N, 5)]
        pairs += [tuple(int(x) for x in rng.integers(0, nd, 2)) for _ in range(40)]
        # add near-diagonal pairs (inside the pattern)
        for _ in range(40):
            i = int(rng.integers(0, nd))
            off = int(rng.integers(-2, 3)) * N * N + int(rng.integers(-2, 3)) * N + int(rng.integers(-2, 3))
            j = min(max(i + off, 0), nd - 1)
            pairs.append((i, j))
        idx = np.array(pairs, dtype=np.uintp)
        out[name + '_idx'] = idx
        out[name + '_multi'] = np.asarray(asm.multi_entries(idx))
        out[name + '_single'] = np.array([asm.entry(int(i), int(j)) for i, j in idx])
    kv2 = bspline.make_knots(3, 0.0, 1.0, 6)
    ann = geometry.quarter_annulus()
    for name, asm in (('stiff2d', assemblers.StiffnessAssembler2D((kv2, kv2), ann)),
                      ('mass2d', assemblers.MassAssembler2D((kv2, kv2), ann))):
        nd = kv2.numdofs ** 2
        rng = np.random.default_rng(11)
        idx = rng.integers(0, nd, (60, 2)).astype(np.uintp)
        near = np.arange(0, nd, 7)
        idx = np.concatenate([idx, np.stack([near, np.minimum(near + 3, nd - 1)], axis=1).astype(np.uintp)])
        out[name + '_idx'] = idx
        out[name + '_multi'] = np.asarray(asm.multi_entries(idx))
    save('entries', **out)


# ---------------------------------------------------------------------------
# (6) Kronecker (geo=None) path and 1D matrices (assemble.py:125-190, 236-282)
def golden_kron():
    out = {}
    kv = bspline.KnotVector(np.array(
        [0., 0., 0., 0., 0., 0.25, 0.35, 0.45, 0.55, 0.65, 0.9, 0.9, 0.9, 0.9, 0.9]), 4)
    out['kv1d'] = kv.kv
    out['M1d'] = assemble.bsp_mass_1d(kv).toarray()
    out['K1d'] = assemble.bsp_stiffness_1d(kv).toarray()
    kvs = (bspline.make_knots(4, 0.0, 1.0, 10), bspline.make_knots(3, 0.0, 1.0, 12))
    put_matrix(out, 'kron2d_stiff', assemble.stiffness(kvs))
    put_matrix(out, 'kron2d_mass', assemble.mass(kvs))
    kvs3 = (bspline.make_knots(3, 0.0, 1.0, 4), bspline.make_knots(3, 0.0, 1.0, 5),
            bspline.make_knots(3, 0.0, 1.0, 6))
    put_matrix(out, 'kron3d_stiff', assemble.stiffness(kvs3))
    put_matrix(out, 'kron3d_mass', assemble.mass(kvs3))
    save('kron', **out)


# ---------------------------------------------------------------------------
# make_knots bit patterns for the BASELINE sizes (bspline.py:192-213, SURVEY A.4 trap 1)
def golden_knots():
    out = {}
    for p, n in ((3, 15), (3, 256), (2, 10), (2, 64), (4, 128), (5, 96), (3, 49), (4, 12), (2, 7)):
        out['p%d_n%d' % (p, n)] = bspline.make_knots(p, 0.0, 1.0, n).kv
    out['p3_n5_m2'] = bspline.make_knots(3, 0.0, 1.0, 5, mult=2).kv
    out['p2_ab'] = bspline.make_knots(2, -1.5, 2.25, 9).kv
    for q in range(1, 8):
        x, w = np.polynomial.legendre.leggauss(q)
        out['leggauss%d_x' % q] = x
        out['leggauss%d_w' % q] = w
    save('knots', **out)


if __name__ == '__main__' and not {'convdiff', 'rhs', 'forms', 'fullsize', 'ondemand', 'vecforms', 'surface', 'pforms'} & set(sys.argv[1:]):
    golden_knots()
    golden_bspline()
    golden_sparsity()
    golden_geometry()
    golden_matrices()
    golden_kron()
    print('reference version', pyiga.__version__)


# ---------------------------------------------------------------------------
# (7) custom form of BASELINE config 5 / SURVEY section 8 f1: convection-diffusion, non-symmetric
#     (assemble.assemble with a run-time compiled vform, pyiga/assemble.py:837-897)
CONVDIFF = '(inner(diff_coeff*grad(u),grad(v)) + inner((x[1],-x[0],1.0),grad(u))*v)*dx'


def golden_convdiff():
    out = {}

    def diff_coeff(x, y, z):
        return 1.0 + x
    cyl = cylinder()
    for name, kvs, geo in (
            ('d3_p2_n3_cyl', (bspline.make_knots(2, 0.0, 1.0, 3),) * 3, cyl),
            ('d3_p32_n243_tbox', (bspline.make_knots(3, 0.0, 1.0, 2), bspline.make_knots(2, 0.0, 1.0, 4),
                                  bspline.make_knots(2, 0.0, 1.0, 3)), geometry.twisted_box()),
            ('d3_mult_cyl', (bspline.make_knots(2, 0.0, 1.0, 3, mult=2), bspline.make_knots(2, 0.0, 1.0, 3),
                             bspline.make_knots(3, 0.0, 1.0, 2)), cyl)):
        A = assemble.assemble(CONVDIFF, kvs, geo=geo, diff_coeff=diff_coeff)
        put_matrix(out, name, A)
        asm = assemble.instantiate_assembler(CONVDIFF, kvs, {'geo': geo, 'diff_coeff': diff_coeff}, None)
        nd = int(np.prod([kv.numdofs for kv in kvs]))
        rng = np.random.default_rng(3)
        idx = rng.integers(0, nd, (50, 2)).astype(np.uintp)
        near = np.arange(0, nd, 5)
        idx = np.concatenate([idx, np.stack([near, np.minimum(near + 2, nd - 1)], 1).astype(np.uintp),
                              np.stack([np.minimum(near + 2, nd - 1), near], 1).astype(np.uintp)])
        out[name + '_idx'] = idx
        out[name + '_multi'] = np.asarray(asm.multi_entries(idx))
    save('convdiff', **out)


if __name__ == '__main__' and 'convdiff' in sys.argv[1:]:
    golden_convdiff()


# ---------------------------------------------------------------------------
# (8) right-hand sides, SURVEY section 8 f3: inner_products / L2Functional assemblers
#     (pyiga/assemble.py:288-340, test/test_assemble.py:223-245,311,428, test/test_solve.py:6-32)
def golden_rhs():
    out = {}

    def f3(x, y, z):
        return np.cos(x) * np.exp(y) * np.sin(z)
    kvs = [bspline.make_knots(p, 0.0, 1.0, 8 + p) for p in range(3, 6)]
    tbox = geometry.twisted_box()
    out['d3_param'] = assemble.inner_products(kvs, f3)
    out['d3_tbox'] = assemble.inner_products(kvs, f3, geo=tbox)
    out['d3_tbox_phys'] = assemble.inner_products(kvs, f3, f_physical=True, geo=tbox)
    out['d3_tbox_asm'] = assemblers.L2FunctionalAssembler3D(kvs, tbox, f=f3).assemble_vector()
    out['d3_tbox_phys_asm'] = assemblers.L2FunctionalAssemblerPhys3D(kvs, tbox, f=f3).assemble_vector()
    cyl = cylinder()
    kv3 = (bspline.make_knots(2, 0.0, 1.0, 5), bspline.make_knots(3, 0.0, 1.0, 4, mult=2), bspline.make_knots(2, 0.0, 1.0, 6))
    out['d3_cyl_phys'] = assemble.inner_products(kv3, f3, f_physical=True, geo=cyl)

    def f2(x, y):
        return np.exp(x + y)

    def fv(x, y):
        return (x * y, x - y)
    kv2 = (bspline.make_knots(3, 0.0, 1.0, 6), bspline.make_knots(2, 0.0, 1.0, 5))
    ann = geometry.quarter_annulus()
    out['d2_param'] = assemble.inner_products(kv2, f2)
    out['d2_ann'] = assemble.inner_products(kv2, f2, geo=ann)
    out['d2_ann_phys'] = assemble.inner_products(kv2, f2, f_physical=True, geo=ann)
    out['d2_ann_vec_phys'] = assemble.inner_products(kv2, fv, f_physical=True, geo=ann)
    out['d2_ann_asm'] = assemblers.L2FunctionalAssembler2D(kv2, ann, f=f2).assemble_vector()
    # spline function as f (parameter domain)
    g = bspline.BSplineFunc(kv2, np.arange(kv2[0].numdofs * kv2[1].numdofs, dtype=float).reshape(kv2[0].numdofs, -1) / 10.0)
    out['d2_splinef'] = assemble.inner_products(kv2, g, geo=ann)
    out['d1_param'] = assemble.inner_products(bspline.make_knots(3, 0.0, 1.0, 7), lambda x: 1 + x ** 2)

    # Poisson problem of test/test_solve.py:6-32 (2D, quarter annulus, Dirichlet data g)
    from pyiga import solvers, approx
    kvs = 2 * (bspline.make_knots(3, 0.0, 1.0, 10),)

    def gfun(x, y):
        return np.cos(x + y) + np.exp(y - x)

    def ffun(x, y):
        return 2 * (np.cos(x + y) - np.exp(y - x))
    bcs = assemble.compute_dirichlet_bcs(kvs, ann, ('all', gfun))
    rhs = assemble.inner_products(kvs, ffun, f_physical=True, geo=ann).ravel()
    A = assemble.stiffness(kvs, geo=ann)
    LS = assemble.RestrictedLinearSystem(A, rhs, bcs)
    u = LS.complete(solvers.make_solver(LS.A, spd=True).dot(LS.b))
    u_ex = approx.project_L2(kvs, gfun, f_physical=True, geo=ann).ravel()
    out['poisson2d_bc_idx'] = np.asarray(bcs[0])
    out['poisson2d_bc_val'] = np.asarray(bcs[1])
    out['poisson2d_rhs'] = rhs
    out['poisson2d_u'] = u
    out['poisson2d_u_ex'] = u_ex
    save('rhs', **out)


if __name__ == '__main__' and 'rhs' in sys.argv[1:]:
    golden_rhs()


# ---------------------------------------------------------------------------
# (9) general form strings: scalar forms that are bilinear in (u, grad u) x (v, grad v),
#     compiled at run time by the reference (pyiga/assemble.py:837-897, pyiga/vform.py:1804-1885)
def form_inputs():
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


def golden_forms():
    out = {}
    inp = form_inputs()
    cyl = cylinder()
    spaces = {
        'cyl_p2': ((bspline.make_knots(2, 0.0, 1.0, 3),) * 3, cyl),
        'tbox_mixed': ((bspline.make_knots(3, 0.0, 1.0, 2), bspline.make_knots(2, 0.0, 1.0, 4, mult=2),
                        bspline.make_knots(1, 0.0, 1.0, 3)), geometry.twisted_box()),
    }
    for sname, (kvs, geo) in spaces.items():
        for fname, (form, names) in FORMS.items():
            A = assemble.assemble(form, kvs, geo=geo, **{k: inp[k] for k in names})
            put_matrix(out, '%s_%s' % (sname, fname), A)
    # 2D forms (quarter annulus, unequal degrees)
    kv2d = (bspline.make_knots(3, 0.0, 1.0, 4), bspline.make_knots(2, 0.0, 1.0, 5, mult=2))
    ann = geometry.quarter_annulus()

    def K2(x, y):
        one = np.ones_like(x * y)
        return np.stack([np.stack(((1.5 + y) * one, 0.4 * x * one), -1), np.stack((-0.3 * one, (2.0 + x * y) * one), -1)], -2)

    def b2(x, y):
        one = np.ones_like(x * y)
        return (y * one, (1.0 - x) * one)
    for fname, form, kw in (('reactdiff', '(inner(grad(u), grad(v)) + c*u*v) * dx', dict(c=lambda x, y: 1.0 + x * y)),
                            ('full', '(inner(dot(K, grad(u)), grad(v)) + inner(b, grad(u)) * v - u * inner(b, grad(v)) + 3 * u * v) * dx', dict(K=K2, b=b2))):
        put_matrix(out, 'd2_%s' % fname, assemble.assemble(form, kv2d, geo=ann, **kw))
    # arity-1 form strings (test/test_assemble.py:426-429)
    kv2 = (bspline.make_knots(3, 0.0, 1.0, 6), bspline.make_knots(2, 0.0, 1.0, 5))
    out['func_d2'] = assemble.assemble('f * v * dx', kv2, geo=geometry.quarter_annulus(), f=lambda x, y: x * y ** 2)
    out['func_d3'] = assemble.assemble('(2 * f + x[0]) * v * dx', spaces['cyl_p2'][0], geo=cyl,
                                       f=lambda x, y, z: np.cos(x) * np.exp(y) * np.sin(z))
    # arity-1 forms with derivatives of v
    out['funcgrad_d2'] = assemble.assemble('(f * v + inner(b, grad(v))) * dx', kv2, geo=geometry.quarter_annulus(),
                                           f=lambda x, y: x * y ** 2, b=b2)
    out['funcgrad_d3'] = assemble.assemble('inner(b, grad(v)) * dx', spaces['tbox_mixed'][0], geo=geometry.twisted_box(), b=inp['b'])
    save('forms', **out)


if __name__ == '__main__' and 'forms' in sys.argv[1:]:
    golden_forms()


# ---------------------------------------------------------------------------
# (10) full-size parity pins (BASELINE configs 2 and 3 at their real sizes; p=4 / p=5 at the largest size the
#      build container holds comfortably): sampled in-pattern entries through multi_entries
#      (pyiga/genericasm.pxi:722-758) and, for config 2, the product of the whole matrix with a fixed vector.
def sample_pairs(ndofs, p, M, seed):
    """M seeded (row, col) pairs inside the sparsity pattern (|i_k - j_k| <= p per axis): random rows plus
    every corner of the dof box, both triangles."""
    rng = np.random.default_rng(seed)
    d = len(ndofs)
    I = np.stack([rng.integers(0, n, M) for n in ndofs], 1)
    corners = np.array(np.meshgrid(*[(0, n - 1) for n in ndofs], indexing='ij')).reshape(d, -1).T
    I[:len(corners)] = corners
    off = rng.integers(-p, p + 1, (M, d))
    J = np.clip(I + off, 0, np.array(ndofs) - 1)
    row = np.ravel_multi_index(I.T, ndofs)
    col = np.ravel_multi_index(J.T, ndofs)
    return np.stack([row, col], 1).astype(np.uintp)


def golden_fullsize():
    out = {}
    ann = geometry.quarter_annulus()
    cyl = cylinder()
    # config 2: 2D p=3 n=256, NURBS quarter annulus, stiffness
    kv = bspline.make_knots(3, 0.0, 1.0, 256)
    kvs = (kv, kv)
    asm = assemblers.StiffnessAssembler2D(kvs, ann)
    idx = sample_pairs([k.numdofs for k in kvs], 3, 20000, 11)
    out['c2_idx'] = idx.astype(np.uint32)
    out['c2_val'] = np.asarray(asm.multi_entries(idx))
    A = assemble.stiffness(kvs, geo=ann)
    x = np.sin(0.37 * np.arange(A.shape[0]) + 0.1)
    out['c2_Ax'] = A @ x
    out['c2_nnz'] = np.array(A.nnz)
    out['c2_absmax'] = np.array(abs(A).max())
    print('c2 done', A.shape, A.nnz)
    # config 3: 3D p=2 n=64, cylinder, mass + stiffness
    kv = bspline.make_knots(2, 0.0, 1.0, 64)
    kvs = (kv, kv, kv)
    idx = sample_pairs([k.numdofs for k in kvs], 2, 10000, 12)
    out['c3_idx'] = idx.astype(np.uint32)
    out['c3_stiff'] = np.asarray(assemblers.StiffnessAssembler3D(kvs, cyl).multi_entries(idx))
    out['c3_mass'] = np.asarray(assemblers.MassAssembler3D(kvs, cyl).multi_entries(idx))
    print('c3 done')
    # config-4 degree on 32^3 spans, config-5 degree on 24^3 spans (the entry sums are the same code path at any n)
    kv = bspline.make_knots(4, 0.0, 1.0, 32)
    kvs = (kv, kv, kv)
    idx = sample_pairs([k.numdofs for k in kvs], 4, 10000, 13)
    out['p4n32_idx'] = idx.astype(np.uint32)
    out['p4n32_stiff'] = np.asarray(assemblers.StiffnessAssembler3D(kvs, cyl).multi_entries(idx))
    print('p4 done')
    kv = bspline.make_knots(5, 0.0, 1.0, 24)
    kvs = (kv, kv, kv)
    idx = sample_pairs([k.numdofs for k in kvs], 5, 6000, 14)
    out['p5n24_idx'] = idx.astype(np.uint32)
    out['p5n24_stiff'] = np.asarray(assemblers.StiffnessAssembler3D(kvs, cyl).multi_entries(idx))

    def diff_coeff(x, y, z):
        return 1.0 + x
    asm = assemble.instantiate_assembler(CONVDIFF, kvs, {'geo': cyl, 'diff_coeff': diff_coeff}, None)
    out['p5n24_convdiff'] = np.asarray(asm.multi_entries(idx))
    print('p5 done')
    save('fullsize', **out)


if __name__ == '__main__' and 'fullsize' in sys.argv[1:]:
    golden_fullsize()


# ---------------------------------------------------------------------------
# (11) on-demand assemblers with a bounding box, SURVEY section 8 f2 (pyiga/codegen/cython.py:541-559, callers
#      pyiga/_hdiscr.py:5-11,37-56): compile_vform(vf, on_demand=True)(kvs, geo, bbox) and _assemble_partial_rows
def golden_ondemand():
    from pyiga import vform, _hdiscr
    from pyiga import compile as pcompile
    out = {}
    cyl = cylinder()
    ann = geometry.quarter_annulus()

    def diff_coeff(x, y, z):
        return 1.0 + x

    def bbox_of(kvs, rows):
        mi = np.unravel_index(rows, [kv.numdofs for kv in kvs])
        box = []
        for kv, ik in zip(kvs, mi):
            s = kv.mesh_support_idx_all()
            box.append((int(s[ik, 0].min()), int(s[ik, 1].max())))
        return tuple(box)

    kvs3 = (bspline.make_knots(2, 0.0, 1.0, 6), bspline.make_knots(3, 0.0, 1.0, 5), bspline.make_knots(2, 0.0, 1.0, 7, mult=2))
    nd3 = [kv.numdofs for kv in kvs3]
    rows3 = np.ravel_multi_index(np.array([(3, 2, 5), (3, 3, 5), (4, 2, 6), (3, 3, 7), (4, 4, 4)]).T, nd3)
    kvs2 = (bspline.make_knots(3, 0.0, 1.0, 9), bspline.make_knots(2, 0.0, 1.0, 12))
    nd2 = [kv.numdofs for kv in kvs2]
    rows2 = np.ravel_multi_index(np.array([(5, 6), (5, 7), (6, 6), (4, 9)]).T, nd2)
    cases = (('stiff3d', vform.stiffness_vf(3), kvs3, {'geo': cyl}, rows3),
             ('mass3d', vform.mass_vf(3), kvs3, {'geo': cyl}, rows3),
             ('convdiff3d', vform.parse_vf(CONVDIFF, kvs3, {'geo': cyl, 'diff_coeff': diff_coeff}), kvs3,
              {'geo': cyl, 'diff_coeff': diff_coeff}, rows3),
             ('stiff2d', vform.stiffness_vf(2), kvs2, {'geo': ann}, rows2),
             ('mass2d', vform.mass_vf(2), kvs2, {'geo': ann}, rows2))
    for name, vf, kvs, args, rows in cases:
        cls = pcompile.compile_vform(vf, on_demand=True)
        bbox = bbox_of(kvs, rows)
        asm = cls(kvs, bbox=bbox, **args)
        A = _hdiscr._assemble_partial_rows(asm, rows).tocsr()
        out[name + '_rows'] = rows.astype(np.int64)
        out[name + '_bbox'] = np.array(bbox, dtype=np.int64)
        sub = A[rows]
        out[name + '_indptr'] = sub.indptr.astype(np.int64)
        out[name + '_indices'] = sub.indices.astype(np.int64)
        out[name + '_data'] = sub.data
        print(name, 'bbox', bbox, 'nnz', sub.nnz)
    save('ondemand', **out)


if __name__ == '__main__' and 'ondemand' in sys.argv[1:]:
    golden_ondemand()


# ---------------------------------------------------------------------------
# (12) form strings with vector-valued basis functions and boundary integrals, SURVEY section 8 f1
#      (test/test_assemble.py:314-400,452-476; pyiga/assemble.py:760-811,837-934)
VEC_FORMS = {
    'nonsym': ('inner(as_matrix([[2,1],[0,0]]).dot(u), v) * dx', [('u', 2), ('v', 2)]),
    'graddiv': ('(inner(grad(u), grad(v)) + div(u) * div(v)) * dx', [('u', 2), ('v', 2)]),
    'divp': ('div(u) * v * dx', [('u', 2), ('v', 1)]),
    'weighted': ('(c * inner(grad(u), grad(v)) + inner(dot(K, u), v)) * dx', [('u', 2), ('v', 2)]),
}
VEC_FORMS3 = {
    'curlcurl': ('(inner(curl(u), curl(v)) + inner(u, v)) * dx', [('u', 3), ('v', 3)]),
}
BD_MATS3 = (('gradgrad_left', 'inner(grad(u), grad(v)) * ds', 'left'), ('gradgrad_top', 'inner(grad(u), grad(v)) * ds', 'top'),
            ('tang_front', 'inner(cross(n, grad(u)), cross(n, grad(v))) * ds', 'front'), ('mass_right', 'u * v * ds', 'right'),
            ('robin_back', '(g * u * v + inner(n, grad(u)) * v) * ds', 'back'))
BD_MATS2 = (('mass_left', 'u * v * ds', 'left'), ('mass_top', 'u * v * ds', 'top'), ('gradgrad_right', 'inner(grad(u), grad(v)) * ds', 'right'),
            ('nitsche_bottom', '(inner(n, grad(u)) * v + g * u * v) * ds', 'bottom'))


def vec_inputs():
    return {'c': lambda x, y: 1.0 + x * y, 'K': lambda x, y: np.stack([np.stack([1.0 + x, 0.5 * y], -1), np.stack([0 * x, 2.0 - y], -1)], -2),
            'f': lambda x, y: x * y ** 2, 'gv': lambda x, y: (y, -x)}


def golden_vecforms():
    out = {}
    ann = geometry.quarter_annulus()
    cyl = cylinder()
    inp = vec_inputs()
    kvs2 = (bspline.make_knots(2, 0.0, 1.0, 5), bspline.make_knots(3, 0.0, 1.0, 4))
    for name, (form, bfuns) in VEC_FORMS.items():
        args = {k: v for k, v in inp.items() if k in form}
        for layout in ('blocked', 'packed'):
            put_matrix(out, '%s_%s' % (name, layout), assemble.assemble(form, kvs2, geo=ann, bfuns=bfuns, layout=layout, **args))
        asm = assemble.instantiate_assembler(form, kvs2, dict(args, geo=ann), bfuns)
        nd = int(np.prod([kv.numdofs for kv in kvs2]))
        idx = np.array([(0, 0), (0, 1), (2, 1), (nd - 1, nd - 2), (nd // 2, nd // 2 + 1), (5, nd - 1)], dtype=np.uintp)
        out[name + '_idx'] = idx
        out[name + '_blocks'] = np.array(asm.multi_blocks(idx))
        out[name + '_numcomp'] = np.array(asm.num_components())
    kvs3 = (bspline.make_knots(2, 0.0, 1.0, 3), bspline.make_knots(2, 0.0, 1.0, 2), bspline.make_knots(3, 0.0, 1.0, 2))
    for name, (form, bfuns) in VEC_FORMS3.items():
        put_matrix(out, name + '_blocked', assemble.assemble(form, kvs3, geo=cyl, bfuns=bfuns, layout='blocked'))
    # arity 1, vector-valued test functions
    for name, form, bfuns in (('fdiv', 'f * div(v) * dx', [('v', 2)]), ('gdotv', 'inner(gv, v) * dx', [('v', 2)])):
        args = {k: v for k, v in inp.items() if k in form}
        for layout in ('blocked', 'packed'):
            out['%s_%s' % (name, layout)] = assemble.assemble(form, kvs2, geo=ann, bfuns=bfuns, layout=layout, **args)
    # boundary integrals, 3D (test/test_assemble.py:331-400)
    kvb = (bspline.make_knots(3, 0.0, 1.0, 3), bspline.make_knots(2, 0.0, 1.0, 4), bspline.make_knots(3, 0.0, 1.0, 5))
    g3 = lambda x, y, z: 1.0 + x + 2 * y * z
    for side in ('left', 'right', 'bottom', 'top', 'front', 'back'):
        out['bd3_v_' + side] = assemble.assemble('v * ds', kvb, geo=cyl, boundary=side)
        out['bd3_gv_' + side] = assemble.assemble('(g * v + inner(n, grad(v))) * ds', kvb, geo=cyl, boundary=side, g=g3)
        out['bd3_vn_' + side] = assemble.assemble('inner(v, n) * ds', kvb, bfuns=[('v', 3)], geo=cyl, boundary=side, layout='packed')
    for name, form, side in BD_MATS3:
        args = {'g': g3} if 'g *' in form else {}
        put_matrix(out, 'bd3_' + name, assemble.assemble(form, kvb, geo=cyl, boundary=side, **args))
    # boundary integrals, 2D
    g2 = lambda x, y: 1.0 + x * y
    for side in ('left', 'right', 'bottom', 'top'):
        out['bd2_v_' + side] = assemble.assemble('g * v * ds', kvs2, geo=ann, boundary=side, g=g2)
        out['bd2_vn_' + side] = assemble.assemble('inner(v, n) * ds', kvs2, bfuns=[('v', 2)], geo=ann, boundary=side, layout='packed')
    for name, form, side in BD_MATS2:
        args = {'g': g2} if 'g *' in form else {}
        put_matrix(out, 'bd2_' + name, assemble.assemble(form, kvs2, geo=ann, boundary=side, **args))
    save('vecforms', **out)


if __name__ == '__main__' and 'vecforms' in sys.argv[1:]:
    golden_vecforms()


# ---------------------------------------------------------------------------
# (13) surface integrals: a patch mapped into a space of one dimension more (test/test_assemble.py:314-330)
def golden_surface():
    out = {}
    cyl = cylinder()
    ann = geometry.quarter_annulus()
    kvs2 = (bspline.make_knots(3, 0.0, 1.0, 4), bspline.make_knots(2, 0.0, 1.0, 6))
    for side in ('left', 'right', 'top', 'back'):
        geo = cyl.boundary(side)
        out['s3_v_' + side] = assemble.assemble('v * ds', kvs2, geo=geo)
        out['s3_xv_' + side] = assemble.assemble('(1.0 + x[0] + 2 * x[1] * x[2]) * v * ds', kvs2, geo=geo)
        out['s3_vn_' + side] = assemble.assemble('inner(v, n) * ds', kvs2, geo=geo, bfuns=[('v', 3)], layout='packed')
        put_matrix(out, 's3_mass_' + side, assemble.assemble('(2.5 + x[0]) * u * v * ds', kvs2, geo=geo))
    kv1 = (bspline.make_knots(3, 0.0, 1.0, 7),)
    for side in ('left', 'right', 'bottom', 'top'):
        geo = ann.boundary(side)
        out['s2_v_' + side] = assemble.assemble('(1.0 + x[0] * x[1]) * v * ds', kv1, geo=geo)
        out['s2_vn_' + side] = assemble.assemble('inner(v, n) * ds', kv1, geo=geo, bfuns=[('v', 2)], layout='packed')
        put_matrix(out, 's2_mass_' + side, assemble.assemble('u * v * ds', kv1, geo=geo))
    save('surface', **out)


if __name__ == '__main__' and 'surface' in sys.argv[1:]:
    golden_surface()


# (14) form strings with second derivatives and parametric derivatives (pyiga/vform.py:1518-1600, transformation of physical
#      Hessians :592-625), and the Hessians of spline / NURBS functions (pyiga/bspline.py:923-975, pyiga/geometry.py:125-150)
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


def golden_pforms():
    out = {}
    inp = form_inputs()
    ann, bann, cyl, tbox = geometry.quarter_annulus(), geometry.bspline_quarter_annulus(), cylinder(), geometry.twisted_box()

    def K2(x, y):
        one = np.ones_like(x * y)
        return np.stack([np.stack(((1.5 + y) * one, 0.4 * x * one), -1), np.stack((-0.3 * one, (2.0 + x * y) * one), -1)], -2)

    def b2(x, y):
        one = np.ones_like(x * y)
        return (y * one, (1.0 - x) * one)
    in2 = dict(c=lambda x, y: 1.0 + x * y, b=b2, K=K2)
    spaces2 = {'ann': ((bspline.make_knots(3, 0.0, 1.0, 4), bspline.make_knots(2, 0.0, 1.0, 5, mult=2)), ann),
               'bann': ((bspline.make_knots(4, 0.0, 1.0, 3), bspline.make_knots(4, 0.0, 1.0, 6)), bann)}
    for sname, (kvs, geo) in spaces2.items():
        for fname, (form, names) in PFORMS2.items():
            put_matrix(out, 'd2_%s_%s' % (sname, fname), assemble.assemble(form, kvs, geo=geo, **{k: in2[k] for k in names}))
    spaces3 = {'cyl_p2': ((bspline.make_knots(2, 0.0, 1.0, 3),) * 3, cyl),
               'tbox_mixed': ((bspline.make_knots(3, 0.0, 1.0, 2), bspline.make_knots(2, 0.0, 1.0, 4, mult=2),
                               bspline.make_knots(2, 0.0, 1.0, 3)), tbox)}
    for sname, (kvs, geo) in spaces3.items():
        for fname, (form, names) in PFORMS3.items():
            put_matrix(out, 'd3_%s_%s' % (sname, fname), assemble.assemble(form, kvs, geo=geo, **{k: inp[k] for k in names}))
    # Hessians of geometry maps on Gauss-like grids
    g2 = (np.linspace(0.02, 0.97, 7), np.linspace(0.05, 0.9, 5))
    g3 = (np.linspace(0.1, 0.9, 3), np.linspace(0.02, 0.97, 4), np.linspace(0.05, 0.9, 5))
    out['hess_grid2_0'], out['hess_grid2_1'] = g2
    out['hess_grid3_0'], out['hess_grid3_1'], out['hess_grid3_2'] = g3
    out['hess_ann'] = ann.grid_hessian(g2)
    out['hess_bann'] = bann.grid_hessian(g2)
    out['hess_cyl'] = cyl.grid_hessian(g3)
    out['hess_tbox'] = tbox.grid_hessian(g3)
    save('pforms', **out)


if __name__ == '__main__' and 'pforms' in sys.argv[1:]:
    golden_pforms()
