"""Drop-in for the tensor-product mass/stiffness path of ``pyiga.assemble``.

Same signatures, return types and assertions as the reference
(pyiga/assemble.py:125-190, 236-282, 703-754, 1009-1049):

    stiffness(kvs, geo=None, format='csr')      mass(kvs, geo=None, format='csr')

With a geometry map the matrix is formed on the MI355X by libigx (assemblers.py); the result
is a scipy sparse matrix with float64 data and int32 indices, canonical (sorted, no
duplicates) and exactly symmetric, like the reference's.  With ``geo=None`` the reference
defines the result as a Kronecker product of 1D matrices built with numpy; that definition
is restated here verbatim in numpy/scipy (only the B-spline evaluation inside goes through
the device).  There is no CPU fallback for the geometry path.
"""
import math

import numpy as np
import scipy.sparse

from . import bspline
from . import assemblers
from . import _lib
from .quadrature import make_iterated_quadrature

################################################################################
# 1D matrices (pyiga/assemble.py:125-190): per-span dense blocks f1 @ (f2*w)^T
################################################################################


def bsp_mixed_deriv_biform_1d(knotvec, du, dv, nqp=None):
    """Matrix of ``a(u,v) = (u^(du), v^(dv))`` for the B-spline basis over `knotvec`.

    Default rule: ``ceil((2p - du - dv + 1)/2)`` Gauss points per span, i.e. p+1 for mass but
    only p for stiffness.
    """
    p, nspans = knotvec.p, knotvec.numspans
    if nqp is None:
        nqp = int(math.ceil((2 * p - du - dv + 1) / 2.0))
    nodes, weights = make_iterated_quadrature(knotvec.mesh, nqp)
    tab = bspline.active_deriv(knotvec, nodes, max(du, dv))        # (nder, p+1, nspans*nqp)
    test = tab[dv].reshape(p + 1, nspans, nqp)                       # rows: test functions
    trial = (tab[du] * weights).reshape(p + 1, nspans, nqp)          # cols: trial functions, weighted
    blocks = np.empty((nspans, p + 1, p + 1))
    for k in range(nspans):
        blocks[k] = np.dot(test[:, k, :], trial[:, k, :].transpose())
    first = knotvec.mesh_span_indices() - p                          # first active dof per span
    loc = np.arange(p + 1)
    rows = (first[:, None, None] + loc[None, :, None]) + 0 * loc[None, None, :]
    cols = (first[:, None, None] + loc[None, None, :]) + 0 * loc[None, :, None]
    n = knotvec.numdofs
    return scipy.sparse.coo_matrix((blocks.ravel(), (rows.ravel(), cols.ravel())), shape=(n, n)).tocsr()


def bsp_mass_1d(knotvec):
    """1D mass matrix."""
    return bsp_mixed_deriv_biform_1d(knotvec, 0, 0)


def bsp_stiffness_1d(knotvec):
    """1D Laplacian stiffness matrix."""
    return bsp_mixed_deriv_biform_1d(knotvec, 1, 1)


################################################################################
# d-dimensional matrices (pyiga/assemble.py:236-282)
################################################################################

_ASSEMBLER = {('mass', 2): assemblers.MassAssembler2D, ('stiffness', 2): assemblers.StiffnessAssembler2D,
              ('mass', 3): assemblers.MassAssembler3D, ('stiffness', 3): assemblers.StiffnessAssembler3D}


def _kron_chain(mats, format):
    out = mats[-1]
    for A in reversed(mats[:-1]):
        out = scipy.sparse.kron(A, out, format=format)
    return out


def _kronecker_form(kind, knotvecs, format):
    """geo=None: the matrix is a (sum of) Kronecker product(s) of 1D matrices."""
    M = [bsp_mass_1d(kv) for kv in knotvecs]
    if kind == 'mass':
        return _kron_chain(M, format)
    K = [bsp_stiffness_1d(kv) for kv in knotvecs]
    total = None
    for a in range(len(knotvecs)):             # derivative on axis a, mass on the others
        term = _kron_chain([K[k] if k == a else M[k] for k in range(len(knotvecs))], format)
        total = term if total is None else total + term
    return total


def separable_terms(kvs, geo, patch3):
    """For a 3D spline map that is separable along axis 0 (``geometry.split_axis0``): the 2D map of the cross-section and
    the weighted 1D matrices of axis 0 as bands ``[ndofs0][2 p0 + 1]`` (row i, column jlo(i) + k) --
    ``m0 = int phi_i phi_j |z'|``, ``k0 = int phi_i' phi_j' / |z'|`` on the Gauss rule of axis 0 of `patch3` -- or None.
    The 3D matrices are then  M = m0 (x) M2D,  K = m0 (x) K2D + k0 (x) M2D  (the Jacobian is block diagonal)."""
    from . import geometry
    sp = geometry.split_axis0(geo)
    if sp is None:
        return None
    zc, geo2 = sp
    kv0, p0 = kvs[0], kvs[0].p
    nodes, w = patch3.gauss(0)
    # z'(xi0) at the Gauss nodes of axis 0: z is a B-spline on geo.kvs[0] with the control values zc
    gkv = geo.kvs[0]
    first_g, vals_g = bspline.collocation_derivs_info(gkv, nodes, derivs=1)
    dz = np.einsum('ga,ga->g', vals_g[1], zc[first_g[:, None] + np.arange(gkv.p + 1)[None, :]])
    if not np.all(np.isfinite(dz)) or np.min(np.abs(dz)) == 0.0:
        return None
    first, vals = bspline.collocation_derivs_info(kv0, nodes, derivs=1)            # (2, G, p+1): values, derivatives
    N0, P = kv0.numdofs, p0 + 1
    supp = kv0.mesh_support_idx_all()
    jlo = np.array([np.searchsorted(supp[:, 1], supp[i, 0], side='right') for i in range(N0)])      # first j whose support meets i's
    C0 = 2 * p0 + 1
    m0, k0 = np.zeros((N0, C0)), np.zeros((N0, C0))
    wm, wk = w * np.abs(dz), w / np.abs(dz)
    for g in range(nodes.shape[0]):
        idx = first[g] + np.arange(P)
        Mg = np.outer(vals[0][g], vals[0][g]) * wm[g]
        Kg = np.outer(vals[1][g], vals[1][g]) * wk[g]
        for a in range(P):
            i = idx[a]
            cols = idx - jlo[i]
            m0[i, cols] += Mg[a]
            k0[i, cols] += Kg[a]
    return geo2, m0, k0


def _separable_form(kind, knotvecs, geo, format):
    """3D mass / stiffness over a geometry that is separable along axis 0: Kronecker expansion on the device (igx_assemble_kron3),
    or None when the map is not of that kind.  OPT-IN (environment IGX_SEPARABLE=1): the default path of ``mass`` / ``stiffness``
    is the general quadrature chain, whose row slabs reproduce it bit for bit; the structured path agrees with it to rounding
    (1e-15 relative) and is ~5x faster at the size of BASELINE config 4."""
    import os
    from . import geometry
    if os.environ.get('IGX_SEPARABLE', '0') != '1' or len(knotvecs) != 3 or geometry.split_axis0(geo) is None:
        return None
    patch3 = assemblers.DevicePatch(knotvecs, geo)
    try:
        terms = separable_terms(knotvecs, geo, patch3)
        if terms is None:
            return None
        geo2, m0, k0 = terms
        patch2 = assemblers.DevicePatch(knotvecs[1:], geo2, nqp=patch3.nqp)      # the Gauss rule of the 3D patch on both axes
        try:
            data = patch3.assemble_kron(kind, patch2, m0, k0)
        except _lib.IgxError as e:
            if e.code != _lib.IGX_ERR_UNSUPPORTED:        # degrees beyond the row buffers of the expansion kernel: the general chain
                raise
            return None
        finally:
            patch2.close()
        indptr, indices = patch3.pattern()
        A = scipy.sparse.csr_matrix((data, indices, indptr), shape=patch3.shape)
        return A if format == 'csr' else A.asformat(format)
    finally:
        patch3.close()


def _tensor_form(kind, knotvecs, geo, format):
    if geo is None:
        return _kronecker_form(kind, knotvecs, format)
    if len(knotvecs) == 3:
        A = _separable_form(kind, knotvecs, geo, format)      # extruded cross-sections: Kronecker products of a 1D and a 2D matrix
        if A is not None:
            return A
    asm = _ASSEMBLER[(kind, len(knotvecs))](knotvecs, geo)
    return assemble_entries(asm, symmetric=True, format=format)


def bsp_mass_2d(knotvecs, geo=None, format='csr'):
    return _tensor_form('mass', tuple(knotvecs), geo, format)


def bsp_stiffness_2d(knotvecs, geo=None, format='csr'):
    return _tensor_form('stiffness', tuple(knotvecs), geo, format)


def bsp_mass_3d(knotvecs, geo=None, format='csr'):
    return _tensor_form('mass', tuple(knotvecs), geo, format)


def bsp_stiffness_3d(knotvecs, geo=None, format='csr'):
    return _tensor_form('stiffness', tuple(knotvecs), geo, format)


################################################################################
# Driver (pyiga/assemble.py:703-754)
################################################################################


def _compute_sparsity_ij(kv_trial, kv_test):
    """All 1D pairs (i, j) -- i test dof, j trial dof -- whose supports share a knot span, in
    row-major order (same set and order as pyiga/mlmatrix.py:420-440).  Supports are the
    mesh-index intervals [lo, hi) of ``mesh_support_idx_all``; because both bounds are
    non-decreasing in the dof index, the partners of row i form one contiguous range."""
    su = kv_trial.mesh_support_idx_all()
    sv = kv_test.mesh_support_idx_all()
    first = np.searchsorted(su[:, 1], sv[:, 0], side='right')      # first j with hi_j > lo_i
    last = np.searchsorted(su[:, 0], sv[:, 1], side='left')        # first j with lo_j >= hi_i
    counts = np.maximum(last - first, 0)
    rows = np.repeat(np.arange(sv.shape[0]), counts)
    cols = np.concatenate([np.arange(a, b) for a, b in zip(first, last)]) if counts.sum() else np.zeros(0, int)
    return np.stack((rows, cols), axis=1).astype(np.uint32)


def _ml_nonzero(kvs0, kvs1, lower_tri):
    """Kronecker expansion of the per-axis pairs in the reference's emission order
    (MLStructure.from_kvs + nonzero: pyiga/mlmatrix.py:59-65,113-130; mlmatrix_cy.pyx:189-289)."""
    I = J = None
    for kv0, kv1 in zip(kvs0, kvs1):
        b = _compute_sparsity_ij(kv0, kv1).astype(np.int64)
        if I is None:
            I, J = b[:, 0], b[:, 1]
        else:
            I = (I[:, None] * kv1.numdofs + b[None, :, 0]).ravel()
            J = (J[:, None] * kv0.numdofs + b[None, :, 1]).ravel()
    if lower_tri:
        keep = J <= I
        I, J = I[keep], J[keep]
    return I, J


def assemble_entries(asm, symmetric=False, format='csr', layout='blocked', algo='auto'):
    """Assemble all entries of the assembler object `asm` into a sparse matrix.

    Device assemblers (``pyiga_amd.assemblers``) form the whole matrix on the GPU; any other
    object with the reference's assembler interface (``arity``, ``kvs``, ``multi_entries``) is
    driven exactly like the reference does (pattern -> multi_entries -> COO -> CSR [+ mirror]).
    """
    from .form_assemblers import FormAssembler
    if isinstance(asm, FormAssembler):
        # vector-valued functions and boundary integrals: one scalar jet form per pair of components on the device
        # (symmetric=True is the reference's request to compute half of the entries; the result is the same matrix)
        if layout not in ('blocked', 'packed'):
            raise ValueError("layout %r: 'blocked' or 'packed'" % (layout,))
        return asm.assemble(format=format, layout=layout)
    if layout != 'blocked':
        raise ValueError("layout %r: only the assemblers of form strings with vector-valued functions know 'packed'" % (layout,))
    if asm.arity == 1:
        return asm.assemble_vector()
    if isinstance(asm, assemblers._DeviceAssembler):
        # mass and stiffness are symmetric forms; the kernels always compute the lower
        # triangle and mirror it, which is what symmetric=True means in the reference
        A = asm.assemble_csr(algo=algo)
        if symmetric and not getattr(asm, '_symmetric_form', True):
            # the reference with symmetric=True computes the lower triangle only and mirrors it, whatever the form
            L = scipy.sparse.tril(A, format='csr')
            A = (L + scipy.sparse.tril(A, k=-1, format='csr').T).tocsr()
        return A.asformat(format)
    kvs0, kvs1 = asm.kvs
    I, J = _ml_nonzero(kvs0, kvs1, lower_tri=symmetric)
    entries = asm.multi_entries(np.column_stack((I, J)))
    shape = (int(np.prod([kv.numdofs for kv in kvs1])), int(np.prod([kv.numdofs for kv in kvs0])))
    A = scipy.sparse.coo_matrix((entries, (I, J)), shape=shape).tocsr()
    if symmetric:
        off = np.nonzero(I != J)[0]
        A += scipy.sparse.coo_matrix((entries[off], (J[off], I[off])), shape=shape)
    return A.asformat(format)


def _nonzeros_for_rows(kvs0, kvs1, row_indices):
    """(I, J) of all pattern entries in the given rows, rows in the given order and columns ascending
    (MLStructure.nonzeros_for_rows, pyiga/mlmatrix.py)."""
    first, last = [], []
    for kv0, kv1 in zip(kvs0, kvs1):
        su, sv = kv0.mesh_support_idx_all(), kv1.mesh_support_idx_all()
        first.append(np.searchsorted(su[:, 1], sv[:, 0], side='right'))
        last.append(np.searchsorted(su[:, 0], sv[:, 1], side='left'))
    nd1 = tuple(kv.numdofs for kv in kvs1)
    nd0 = tuple(kv.numdofs for kv in kvs0)
    I, J = [], []
    for r in np.asarray(row_indices, dtype=np.int64).ravel():
        mi = np.unravel_index(r, nd1)
        cols = np.zeros(1, dtype=np.int64)
        for k, ik in enumerate(mi):
            cols = (cols[:, None] * nd0[k] + np.arange(first[k][ik], last[k][ik])[None, :]).ravel()
        I.append(np.full(cols.shape[0], r, dtype=np.int64))
        J.append(cols)
    if not I:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(I), np.concatenate(J)


def bbox_for_rows(kvs, row_indices):
    """Cell ranges ((lo, hi) per axis, upper limit exclusive) that contain the supports of the given basis functions
    (ravelled dof indices) -- the bounding box the hierarchical discretisation passes to an on-demand assembler
    (pyiga/_hdiscr.py:225-233)."""
    nd = tuple(kv.numdofs for kv in kvs)
    mi = np.unravel_index(np.asarray(row_indices, dtype=np.int64).ravel(), nd)
    if mi[0].size == 0:
        return tuple((0, 0) for _ in kvs)
    box = []
    for kv, ik in zip(kvs, mi):
        supp = kv.mesh_support_idx_all()
        box.append((int(supp[ik, 0].min()), int(supp[ik, 1].max())))
    return tuple(box)


def assemble_partial_rows(asm, row_indices):
    """Submatrix (full shape, CSR) that contains only the given rows -- what the hierarchical
    discretisation asks of an assembler (pyiga/_hdiscr.py:5-11): the pattern entries of those rows go
    through ``asm.multi_entries`` in one batch (on the device for the assemblers of this package)."""
    kvs0, kvs1 = asm.kvs
    I, J = _nonzeros_for_rows(kvs0, kvs1, row_indices)
    data = asm.multi_entries(np.column_stack((I, J)).astype(np.uintp)) if I.size else np.zeros(0)
    shape = (int(np.prod([kv.numdofs for kv in kvs1])), int(np.prod([kv.numdofs for kv in kvs0])))
    return scipy.sparse.coo_matrix((data, (I, J)), shape=shape).tocsr()


################################################################################
# Right-hand sides (pyiga/assemble.py:288-340)
################################################################################

def inner_products(kvs, f, f_physical=False, geo=None):
    """L2 inner products of every basis function of the tensor product basis with `f` (the load
    vector); array of shape ``ndofs[0] x ... x ndofs[-1]`` (+ the component axes of a non-scalar `f`).

    `f` is a function of (x, y, z) or a spline function object; `f_physical` says whether it is given
    in physical coordinates (then `geo` is required).  Without `geo` the integrals are over the
    parameter domain.  2D and 3D run on the device; 1D is a few numpy lines on the host.
    """
    from . import geometry, utils
    from .quadrature import make_tensor_quadrature
    if isinstance(kvs, bspline.KnotVector):
        kvs = (kvs,)
    kvs = tuple(kvs)
    dim = len(kvs)
    if f_physical:
        assert geo is not None, 'inner_products in physical domain requires geometry'
    if dim == 1:
        # same steps as the reference: weights, transposed collocation matrix (host; 1D is outside the device path)
        grid, gw = make_tensor_quadrature([kvs[0].mesh], kvs[0].p + 1)
        fvals = np.array(utils.grid_eval_transformed(f, grid, geo) if f_physical else utils.grid_eval(f, grid), dtype=float)
        w = gw[0].copy()
        if geo is not None:
            # a curve: |det J| of the reference's `determinants` is |x'(t)| for a scalar map (pyiga/assemble.py:326-333)
            jac = np.asarray(geo.grid_jacobian(grid), dtype=float).reshape(len(grid[0]), -1)
            assert jac.shape[1] == 1, '1D inner products need a scalar-valued geometry map'
            w = w * np.abs(jac[:, 0])
        w = w.reshape((-1,) + (1,) * (fvals.ndim - 1))
        return bspline.collocation(kvs[0], grid[0]).T @ (fvals * w)
    assert dim in (2, 3), 'Dimensions higher than 3 are currently not implemented.'
    g = geo if geo is not None else geometry.unit_cube(dim)     # parameter domain: |det J| = 1
    patch = assemblers.DevicePatch(kvs, g)
    # a plain scalar callable is traced into a C expression and evaluated at the Gauss points on the device
    # together with geometry and weight by ONE generated kernel (pyiga_amd.symbolic, igx_load_vector_expr): no sampling on the
    # host, no upload, the products W f are the only full-grid array; anything else as in the reference
    from . import symbolic
    src = symbolic.trace_function(f, dim) if (not f_physical or isinstance(geo, (bspline.BSplineFunc, geometry.NurbsFunc))) else None
    out = None
    if src is not None:
        try:
            out = patch.load_vector_expr(src, parametric=not f_physical)
        except _lib.IgxError as e:                               # (no run-time compiler on this box / not traceable into a kernel: sampled on the host)
            if not _lib.sampled_fallback(e, 'the function of a load vector'):
                raise
            out = None
    if out is None:
        grid = tuple(patch.gauss(k)[0] for k in range(dim))
        fvals = utils.grid_eval_transformed(f, grid, geo) if f_physical else utils.grid_eval(f, grid)
        out = patch.load_vector(fvals)
    patch.close()
    return out


################################################################################
# Custom forms (pyiga/assemble.py:837-897)
################################################################################

def _normalise_form(problem):
    return ''.join(str(problem).split())


# variational forms with a hand-written device implementation, keyed by their whitespace-free text
_KNOWN_FORMS = {
    'inner(grad(u),grad(v))*dx': 'stiffness',
    'u*v*dx': 'mass',
    '(inner(diff_coeff*grad(u),grad(v))+inner((x[1],-x[0],1.0),grad(u))*v)*dx': 'convdiff',
}


def instantiate_assembler(problem, kvs, args, bfuns=None, boundary=None):
    """Assembler object for `problem`.  The reference compiles arbitrary form strings at run time
    (vform -> Cython -> gcc).  Here the three built-in forms map to their hand-written kernels, any other
    3D scalar form in u, v, grad, inner, dot goes through the general device form (``pyiga_amd.forms`` ->
    ``IGX_FORM``); assembler classes and objects are accepted as in the reference."""
    surface = isinstance(problem, str) and 'geo' in args and getattr(args['geo'], 'dim', None) == len(tuple(kvs)) + 1
    if isinstance(problem, str) and (bfuns is not None or boundary is not None or surface):
        # vector-valued basis functions and/or a boundary integral (pyiga/vform.py:1822-1845, pyiga/assemble.py:929-934)
        from .form_assemblers import FormAssembler
        if 'geo' not in args:
            raise ValueError("required input parameter 'geo' missing")
        return FormAssembler(tuple(kvs), args['geo'], problem, bfuns=bfuns, inputs=args, boundary=boundary)
    assert bfuns is None and boundary is None, 'custom basis functions / boundary forms need a form string'
    if isinstance(problem, str):
        kind = _KNOWN_FORMS.get(_normalise_form(problem))
        if 'geo' not in args:
            raise ValueError("required input parameter 'geo' missing")
        kvs = tuple(kvs)
        from . import forms
        if kind is None and forms.arity(problem) == 1:
            if len(kvs) not in (2, 3):
                raise NotImplementedError('linear functionals are supported for 2D and 3D patches')
            cls = assemblers.GeneralFunctionalAssembler2D if len(kvs) == 2 else assemblers.GeneralFunctionalAssembler3D
            return cls(kvs, args['geo'], problem, inputs=args)
        if kind is None:
            # any other scalar form that is bilinear in (u, grad u) x (v, grad v): general device form
            if len(kvs) not in (2, 3):
                raise NotImplementedError('general form strings are supported for 2D and 3D patches')
            cls = assemblers.GeneralFormAssembler2D if len(kvs) == 2 else assemblers.GeneralFormAssembler3D
            try:
                return cls(kvs, args['geo'], problem, inputs=args)
            except NotImplementedError:
                # second derivatives / parametric derivatives: parametric jet form assembled in passes (pyiga_amd.pforms)
                cls = assemblers.ParametricFormAssembler2D if len(kvs) == 2 else assemblers.ParametricFormAssembler3D
                return cls(kvs, args['geo'], problem, inputs=args)
        if kind == 'convdiff':
            if 'diff_coeff' not in args:
                raise ValueError("required input parameter 'diff_coeff' missing")
            assert len(kvs) == 3, 'the convection-diffusion form is three-dimensional'
            return assemblers.ConvDiffAssembler3D(kvs, args['geo'], args['diff_coeff'])
        return _ASSEMBLER[(kind, len(kvs))](kvs, args['geo'])
    if isinstance(problem, type):
        used = {k: args[k] for k in problem.inputs() if k in args}
        missing = [k for k in problem.inputs() if k not in args]
        if missing:
            raise ValueError("required input parameter '%s' missing" % missing[0])
        return problem(tuple(kvs), **used)
    return problem          # already an assembler object


def assemble(problem, kvs, args=None, bfuns=None, boundary=None, symmetric=False, format='csr', layout='blocked', **kwargs):
    """Assemble the matrix of a variational form (string, assembler class or assembler object); signature of
    pyiga/assemble.py:837.  `bfuns` (vector-valued basis functions), `boundary` (a face of the patch, forms in ``ds``) and
    `layout` ('blocked' | 'packed', vector-valued spaces) as in the reference; they go through
    ``pyiga_amd.form_assemblers.FormAssembler``."""
    args = dict(args or {})
    args.update(kwargs)
    asm = instantiate_assembler(problem, kvs, args, bfuns, boundary)
    return assemble_entries(asm, symmetric=symmetric, format=format, layout=layout)


class Assembler:
    """Assembler object for a sequence of problems with changing inputs (interface of pyiga/assemble.py:958-1003):
    ``Assembler(problem, kvs, geo=..., updatable=['geo'])``, then ``assemble(geo=new_geo)`` or ``update(geo=new_geo)`` +
    ``assemble()``.  `updatable` names must be inputs of the problem (``ValueError``), only they may be updated
    (``RuntimeError``).  The device assembler is instantiated again when an input changes: the per-patch set-up is a few
    milliseconds, everything geometry-dependent has to be recomputed anyway."""

    def __init__(self, problem, kvs, args=None, bfuns=None, boundary=None, symmetric=False, updatable=(), **kwargs):
        self._args = dict(args or {})
        self._args.update(kwargs)
        self._problem, self._kvs, self._bfuns, self._boundary = problem, kvs, bfuns, boundary
        self.symmetric = bool(symmetric)
        self.updatable = tuple(updatable)
        self.asm = instantiate_assembler(problem, kvs, self._args, bfuns, boundary)
        inputs = self._input_names()
        if not all(name in inputs for name in self.updatable):
            raise ValueError('Assembler received an updatable argument which is not an assembler input')

    def _input_names(self):
        if isinstance(self._problem, str):
            import re
            words = set(re.findall(r"[^\d\W]\w*", self._problem))
            return {'geo'} | (words & set(self._args))
        return set(self.asm.inputs().keys()) if hasattr(self.asm, 'inputs') else {'geo'}

    def update(self, **kwargs):
        if not all(name in self.updatable for name in kwargs):
            raise RuntimeError('update() received an argument which was not specified as updatable')
        self._args.update(kwargs)
        self.asm = instantiate_assembler(self._problem, self._kvs, self._args, self._bfuns, self._boundary)

    def assemble(self, format='csr', layout='blocked', **upd_fields):
        if upd_fields:
            self.update(**upd_fields)
        return assemble_entries(self.asm, symmetric=self.symmetric, format=format, layout=layout)


################################################################################
# Convenience functions (pyiga/assemble.py:1009-1049)
################################################################################


def _detect_dim(kvs):
    if isinstance(kvs, bspline.KnotVector):
        return 1, kvs
    d = len(kvs)
    return d, (kvs[0] if d == 1 else kvs)


def _convenience(kind, kvs, geo, format):
    dim, kvs = _detect_dim(kvs)
    if geo:
        assert geo.dim == dim, 'Geometry has wrong dimension'
    if dim == 1:
        assert geo is None, 'Geometry map not supported for 1D assembling'
        return bsp_mass_1d(kvs) if kind == 'mass' else bsp_stiffness_1d(kvs)
    assert dim in (2, 3), 'Dimensions higher than 3 are currently not implemented.'
    return _tensor_form(kind, tuple(kvs), geo, format)


def mass(kvs, geo=None, format='csr'):
    """Mass matrix for a (tensor product) B-spline basis with an optional geometry map."""
    return _convenience('mass', kvs, geo, format)


def stiffness(kvs, geo=None, format='csr'):
    """Stiffness matrix for a (tensor product) B-spline basis with an optional geometry map."""
    return _convenience('stiffness', kvs, geo, format)


################################################################################
# Low-rank "fast" variants (pyiga/assemble.py:1063-1101)
################################################################################

def _fast(kind, kvs, geo, tol, maxiter, skipcount, tolcount, verbose):
    if geo is None:
        # the default assemblers use Kronecker product assembling if no geometry is present (pyiga/assemble.py:1068-1070)
        return mass(kvs) if kind == 'mass' else stiffness(kvs)
    dim, kvs = _detect_dim(kvs)
    assert geo.dim == dim, 'Geometry has wrong dimension'
    assert dim != 1, 'Geometry map not supported for 1D assembling'
    assert dim in (2, 3), 'Dimensions higher than 3 are currently not implemented.'
    patch = assemblers.DevicePatch(tuple(kvs), geo)
    try:
        return patch.fast_assemble(kind, tol=tol, maxiter=maxiter, skipcount=skipcount, tolcount=tolcount, verbose=verbose)
    finally:
        patch.close()


def mass_fast(kvs, geo=None, tol=1e-10, maxiter=100, skipcount=3, tolcount=3, verbose=2):
    """Mass matrix by the low-rank (adaptive cross approximation) assembler (pyiga/assemble.py:1063-1081,
    pyiga/fastasm.cc): the ACA control flow runs on the host, rows / columns / fibres of the reordered matrix are
    evaluated on the device in batches.  Approximate to `tol`; on this hardware the exact assembly
    (:func:`mass`) is faster -- the entry point exists for code written against pyiga."""
    return _fast('mass', kvs, geo, tol, maxiter, skipcount, tolcount, verbose)


def stiffness_fast(kvs, geo=None, tol=1e-10, maxiter=100, skipcount=3, tolcount=3, verbose=2):
    """See :func:`mass_fast` (pyiga/assemble.py:1083-1101)."""
    return _fast('stiffness', kvs, geo, tol, maxiter, skipcount, tolcount, verbose)
