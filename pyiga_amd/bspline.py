"""Knot vectors, B-spline evaluation and spline functions -- host mirror of the parts
of ``pyiga.bspline`` the assembly path needs (pyiga/bspline.py:36-222,591-660,820-921).

Integer bookkeeping (meshes, supports) is numpy on the host, as in the reference.
Everything that evaluates B-splines (``active_deriv``, ``grid_eval``, ``grid_jacobian``)
runs on the MI355X through libigx; there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib


class KnotVector:
    """Open B-spline knot vector with a spline degree (interface of pyiga/bspline.py:36-190).

    The knots are run-length coded once, when the vector is built: ``mesh`` holds the distinct knots (the breakpoints)
    and ``_cell`` the index of the breakpoint each knot sits on.  Supports, nonempty spans and mesh supports are
    slices of these two arrays.

    Attributes:
        kv (ndarray): the knots; first and last repeated ``p+1`` times
        p (int): spline degree
    """

    def __init__(self, knots, p):
        kv = np.asarray(knots)
        rises = np.diff(kv)
        assert not (rises < 0).any(), 'knots should be increasing'
        self.kv = kv
        self.p = p
        starts = np.concatenate(([True], rises > 0)) if kv.size else np.zeros(0, dtype=bool)
        self._breaks = kv[starts]                          # distinct knots, in order
        self._cell = np.cumsum(starts) - 1                 # knot index -> index into the breakpoints
        self._rising = np.flatnonzero(rises > 0)           # knot indices i with kv[i] < kv[i+1]

    # ---- sizes
    @property
    def numknots(self):
        return self.kv.size

    @property
    def numdofs(self):
        """Number of B-splines over this knot vector."""
        return self.kv.size - (self.p + 1)

    @property
    def numspans(self):
        """Number of nonempty knot intervals."""
        return self._breaks.size - 1

    @property
    def mesh(self):
        """The distinct knots."""
        return self._breaks

    def meshsize_avg(self):
        return abs(self.kv[-1] - self.kv[0]) / self.numspans

    # ---- supports: B-spline j lives on the knots j .. j+p+1
    def support_idx(self, j):
        return (j, j + self.p + 1)

    def support(self, j=None):
        lo, hi = (0, self.kv.size - 1) if j is None else self.support_idx(j)
        return (self.kv[lo], self.kv[hi])

    def mesh_support_idx(self, j):
        lo, hi = self.support_idx(j)
        return (self._cell[lo], self._cell[hi])

    def mesh_support_idx_all(self):
        """``N x 2`` array: first and one-past-last mesh index of every B-spline's support."""
        n, width = self.numdofs, self.p + 1
        return np.column_stack((self._cell[:n], self._cell[width:width + n]))

    def mesh_span_indices(self):
        """Knot indices ``i`` with ``kv[i] != kv[i+1]`` (one per nonempty span)."""
        return self._rising

    # ---- spans
    def findspan(self, u):
        """Index ``i`` with ``kv[i] <= u < kv[i+1]`` (last span closed on the right)."""
        return int(findspans(self, np.array([u], dtype=float))[0])

    def first_active(self, k):
        return k - self.p

    def first_active_at(self, u):
        return self.first_active(self.findspan(u))

    # ---- value semantics
    def copy(self):
        return KnotVector(self.kv.copy(), self.p)

    def __eq__(self, other):
        if self.p != other.p or self.kv.shape != other.kv.shape:
            return False
        return bool(np.allclose(self.kv, other.kv, atol=1e-8, rtol=1e-8))

    def __str__(self):
        return '<KnotVector p=%d sz=%d>' % (self.p, self.kv.size)

    def __repr__(self):
        return 'KnotVector(%r, %r)' % (self.kv, self.p)


def make_knots(p, a, b, n, mult=1):
    """Open knot vector of degree `p` over `(a,b)` with `n` spans (pyiga/bspline.py:192-213).

    Uses the same ``np.arange`` expression as the reference so the knots agree bit for bit
    (including its quirk of producing ``n+1`` spans for a few ``n``).
    """
    kv = np.concatenate(
        (np.repeat(a, p + 1),
         np.repeat(np.arange(a, b, (b - a) / n)[1:], mult),
         np.repeat(b, p + 1)))
    return KnotVector(kv, p)


def numdofs(kvs):
    if isinstance(kvs, KnotVector):
        return kvs.numdofs
    return np.prod([kv.numdofs for kv in kvs])


# ---------------------------------------------------------------------------------------------
# evaluation on the device (pyiga/bspline_cy.pyx)
def active_deriv(knotvec, u, numderiv):
    """All active B-splines and their derivatives up to `numderiv` at the points `u`.

    Shape ``(numderiv+1, p+1)`` for scalar `u`, else ``(numderiv+1, p+1, len(u))``
    (pyiga/bspline_cy.pyx:126-145).
    """
    scalar = np.isscalar(u)
    uu = _lib.f64(np.atleast_1d(u))
    kv = _lib.f64(knotvec.kv)
    out = np.empty((numderiv + 1, knotvec.p + 1, uu.shape[0]))
    ctx = _lib.context()
    _lib.check(_lib.load().igx_active_deriv(ctx.handle, _lib.dptr(kv), kv.size, int(knotvec.p),
                                            _lib.dptr(uu), uu.shape[0], int(numderiv), _lib.dptr(out)),
               'igx_active_deriv')
    return out[:, :, 0] if scalar else out


def active_ev(knotvec, u):
    """Values of the active B-splines, shape ``(p+1, len(u))``."""
    return active_deriv(knotvec, u, 0)[0]


def findspans(knotvec, u):
    """Vectorised ``pyx_findspan`` (pyiga/bspline_cy.pyx:13-38)."""
    uu = _lib.f64(np.atleast_1d(u))
    kv = _lib.f64(knotvec.kv)
    out = np.empty(uu.shape[0], dtype=np.int64)
    ctx = _lib.context()
    _lib.check(_lib.load().igx_find_spans(ctx.handle, _lib.dptr(kv), kv.size, int(knotvec.p), _lib.dptr(uu),
                                          uu.shape[0], out.ctypes.data_as(C.POINTER(C.c_int64))),
               'igx_find_spans')
    return out


def collocation_derivs_info(kv, nodes, derivs=1):
    """First active index per node and ``(derivs+1) x len(nodes) x (p+1)`` coefficients
    (pyiga/bspline.py:648-660)."""
    nodes = _lib.f64(nodes)
    values = active_deriv(kv, nodes, derivs)
    indices = findspans(kv, nodes) - kv.p
    return indices, values.swapaxes(-2, -1)


def collocation(kv, nodes):
    """Sparse collocation matrix ``len(nodes) x numdofs`` of the B-spline basis (pyiga/bspline.py:629-646)."""
    import scipy.sparse
    nodes = _lib.f64(nodes)
    first, values = collocation_derivs_info(kv, nodes, derivs=0)
    m, P = nodes.shape[0], kv.p + 1
    indices = (first[:, None] + np.arange(P)[None, :]).ravel()
    indptr = np.arange(0, m * P + 1, P)
    return scipy.sparse.csr_matrix((values[0].ravel(), indices, indptr), shape=(m, kv.numdofs))


def _derivative_matrices(kv, nodes, maxorder):
    """Dense ``len(nodes) x numdofs`` matrices of the basis and its derivatives up to `maxorder` at `nodes`."""
    nodes = _lib.f64(nodes)
    first, vals = collocation_derivs_info(kv, nodes, derivs=maxorder)      # (maxorder+1) x len(nodes) x (p+1)
    cols = first[:, None] + np.arange(kv.p + 1)[None, :]
    rows = np.broadcast_to(np.arange(nodes.shape[0])[:, None], cols.shape)
    mats = np.zeros((maxorder + 1, nodes.shape[0], kv.numdofs))
    for r in range(maxorder + 1):
        mats[r][rows, cols] = vals[r]
    return mats


def tensor_partials(kvs, coeffs, gridaxes, maxorder=2):
    """All partial derivatives up to total order `maxorder` of a tensor-product spline on a tensor grid:
    dict ``orders -> array``, `orders` a tuple with one derivative order per GRID axis (the order of `kvs`), arrays of shape
    ``shape(grid) + coeffs.shape[sdim:]``.  The 1D derivative values come from the device (``active_deriv``), the
    contractions over the axes run in numpy: a set-up helper for the geometry Hessians of forms with second derivatives
    (pyiga/bspline.py:923-975 ``grid_hessian``), sized for the patches those forms are assembled on."""
    import itertools
    sdim = len(kvs)
    mats = [_derivative_matrices(kv, ax, maxorder) for kv, ax in zip(kvs, gridaxes)]
    out = {}
    for orders in itertools.product(range(maxorder + 1), repeat=sdim):
        if sum(orders) > maxorder:
            continue
        vals = np.asarray(coeffs, dtype=float)
        for a in range(sdim):
            vals = np.moveaxis(np.tensordot(mats[a][orders[a]], vals, axes=(1, a)), 0, a)
        out[orders] = vals
    return out


def _hessian_index_pairs(sdim):
    """(i, j), i <= j, in the order of the linearised Hessians: (xx, xy, yy) resp. (xx, xy, xz, yy, yz, zz)."""
    return [(i, j) for i in range(sdim) for j in range(i, sdim)]


def _orders_of(sdim, *xyz):
    """Derivative orders per grid axis for derivatives along the coordinates `xyz` (x = LAST grid axis)."""
    o = [0] * sdim
    for k in xyz:
        o[sdim - 1 - k] += 1
    return tuple(o)


# ---------------------------------------------------------------------------------------------
def _geo_desc(kvs, coeffs, nurbs):
    """Fill the geometry part of an igx_patch_desc; returns (desc, keepalive)."""
    d = _lib.PatchDesc()
    sdim = len(kvs)
    d.dim = sdim
    d.geo_kind = _lib.IGX_GEO_NURBS if nurbs else _lib.IGX_GEO_BSPLINE
    keep = []
    for k, kv in enumerate(kvs):
        a = _lib.f64(kv.kv)
        keep.append(a)
        d.geo_kv[k] = _lib.dptr(a)
        d.geo_kv_len[k] = a.size
        d.geo_p[k] = int(kv.p)
    c = _lib.f64(coeffs)
    keep.append(c)
    d.ctrl = _lib.dptr(c)
    return d, keep


def _device_grid_eval(kvs, coeffs, nurbs, ncomp, gridaxes, want_jac):
    sdim = len(kvs)
    if sdim == 1:
        # curves: basis values and derivatives from the device (k_basis_tables), the contraction with the control
        # points is a few numpy lines (1D is outside the device path, like the reference's 1D assembling)
        kv = kvs[0]
        nodes = _lib.f64(np.squeeze(gridaxes[0]) if np.ndim(gridaxes[0]) != 1 else gridaxes[0])
        first, val = collocation_derivs_info(kv, nodes, derivs=1)
        idx = first[:, None] + np.arange(kv.p + 1)[None, :]
        Cn = np.asarray(coeffs, dtype=float).reshape(kv.numdofs, -1)[idx]          # (m, P, ncomp [+ weight])
        hv = np.einsum('mp,mpc->mc', val[0], Cn)
        hd = np.einsum('mp,mpc->mc', val[1], Cn)
        if nurbs:
            W, dW = hv[:, -1:], hd[:, -1:]
            ev = hv[:, :-1] / W
            jac = (hd[:, :-1] * W - hv[:, :-1] * dW) / (W * W)                       # quotient rule, pyiga/geometry.py:17-25
        else:
            ev, jac = hv, hd
        return jac[..., None] if want_jac else ev
    if sdim not in (2, 3):
        raise NotImplementedError('device spline evaluation supports 1D, 2D and 3D parameter domains')
    d, keep = _geo_desc(kvs, coeffs, nurbs)
    axes = [_lib.f64(np.squeeze(ax) if np.ndim(ax) != 1 else ax) for ax in gridaxes]
    assert all(ax.ndim == 1 for ax in axes), 'Grid axes should be one-dimensional'
    grid = (_lib._dp * 3)()
    ng = (C.c_int32 * 3)()
    for k, ax in enumerate(axes):
        grid[k] = _lib.dptr(ax)
        ng[k] = ax.shape[0]
    shape = tuple(ax.shape[0] for ax in axes)
    jac = np.empty(shape + (ncomp, sdim)) if want_jac else None
    ev = np.empty(shape + (ncomp,)) if not want_jac else None
    ctx = _lib.context()
    _lib.check(_lib.load().igx_grid_jacobian(ctx.handle, C.byref(d), int(ncomp), grid, ng,
                                             _lib.dptr(jac) if want_jac else None,
                                             _lib.dptr(ev) if not want_jac else None),
               'igx_grid_jacobian')
    return jac if want_jac else ev


class _BaseSplineFunc:
    def is_scalar(self):
        return len(self.output_shape()) == 0

    def is_vector(self):
        return len(self.output_shape()) == 1

    @property
    def support(self):
        return tuple(kv.support() for kv in self.kvs)


class BSplineFunc(_BaseSplineFunc):
    """Function given by tensor-product B-spline coefficients (pyiga/bspline.py:820-921).

    `kvs` are in (z, y, x) order; trailing axes of `coeffs` give the output dimension.
    """

    def __init__(self, kvs, coeffs):
        if isinstance(kvs, KnotVector):
            kvs = (kvs,)
        self.kvs = tuple(kvs)
        self.sdim = len(kvs)
        N = tuple(kv.numdofs for kv in kvs)
        coeffs = np.asanyarray(coeffs)
        if coeffs.ndim == 1:
            assert coeffs.shape[0] == np.prod(N), 'Wrong length of coefficient vector'
            coeffs = coeffs.reshape(N)
        assert N == coeffs.shape[:self.sdim], 'Wrong shape of coefficients'
        self.coeffs = coeffs
        dim = coeffs.shape[self.sdim:]
        if len(dim) == 0:
            dim = 1
        elif len(dim) == 1:
            dim = dim[0]
        self.dim = dim

    def output_shape(self):
        return self.coeffs.shape[self.sdim:]

    def copy(self):
        return BSplineFunc(tuple(kv.copy() for kv in self.kvs), self.coeffs.copy())

    def _ncomp(self):
        assert len(self.output_shape()) <= 1, 'tensor-valued functions are not supported on the device'
        return int(np.prod(self.output_shape())) if self.output_shape() else 1

    def grid_eval(self, gridaxes):
        """Values on a tensor grid (x axis last)."""
        assert len(gridaxes) == self.sdim, 'Input has wrong dimension'
        nc = self._ncomp()
        out = _device_grid_eval(self.kvs, self.coeffs.reshape(self.coeffs.shape[:self.sdim] + (nc,)),
                                False, nc, gridaxes, want_jac=False)
        return out[..., 0] if self.is_scalar() else out

    def grid_jacobian(self, gridaxes):
        """Jacobians ``shape(grid) x dim x sdim``; last axis is d/d(x,y,z)."""
        assert len(gridaxes) == self.sdim, 'Input has wrong dimension'
        nc = self._ncomp()
        out = _device_grid_eval(self.kvs, self.coeffs.reshape(self.coeffs.shape[:self.sdim] + (nc,)),
                                False, nc, gridaxes, want_jac=True)
        return out[..., 0, :] if self.is_scalar() else out

    def grid_hessian(self, gridaxes):
        """Second derivatives ``shape(grid) x dim x num_hess`` (the `dim` axis is dropped for scalar functions), the
        symmetric part linearised as (xx, xy, yy) resp. (xx, xy, xz, yy, yz, zz), x = last grid axis
        (pyiga/bspline.py:923-975)."""
        assert len(gridaxes) == self.sdim, 'Input has wrong dimension'
        nc = self._ncomp()
        part = tensor_partials(self.kvs, self.coeffs.reshape(self.coeffs.shape[:self.sdim] + (nc,)), gridaxes, 2)
        H = np.stack([part[_orders_of(self.sdim, i, j)] for i, j in _hessian_index_pairs(self.sdim)], axis=-1)
        return H[..., 0, :] if self.is_scalar() else H

    def boundary(self, bdspec):
        """One face of the parameter domain as a spline function with `sdim` reduced by one (pyiga/bspline.py:1014-1036):
        with open knot vectors the face's control net is the first / last layer of the net along that axis."""
        from .form_assemblers import parse_bdspec
        axis, side = parse_bdspec(bdspec, self.sdim)
        layer = np.take(self.coeffs, 0 if side == 0 else -1, axis=axis)
        return BSplineFunc(self.kvs[:axis] + self.kvs[axis + 1:], layer)

    def as_nurbs(self):
        from .geometry import NurbsFunc
        return NurbsFunc(self.kvs, self.coeffs.copy(), np.ones(self.coeffs.shape[:self.sdim]))

    def as_vector(self):
        if self.is_vector():
            return self
        assert self.is_scalar()
        return BSplineFunc(self.kvs, self.coeffs[..., None])
