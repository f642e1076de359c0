"""Tensor-product B-spline / NURBS geometry maps -- host mirror of the parts of
``pyiga.geometry`` the assembly path uses (pyiga/geometry.py:17-123,425-491,533-615,755-809).

Control nets live on the host (they are tiny); evaluation on tensor grids runs on the
device through libigx.  When a geometry is handed to ``assemble.mass/stiffness`` it is never
evaluated on the host at all: the kernels take the control net directly.
"""
import functools

import numpy as np

from . import bspline
from .bspline import BSplineFunc, _BaseSplineFunc, _device_grid_eval


class NurbsFunc(_BaseSplineFunc):
    """Function in a tensor-product NURBS basis (interface of pyiga/geometry.py:27-123).

    Stored homogeneously: ``coeffs[..., :-1]`` are the coefficients multiplied by their weights, ``coeffs[..., -1]``
    the weights -- the control net the device kernels take as it is.
    """

    def __init__(self, kvs, coeffs, weights, premultiplied=False):
        self.kvs = (kvs,) if isinstance(kvs, bspline.KnotVector) else tuple(kvs)
        self.sdim = len(self.kvs)
        grid = tuple(kv.numdofs for kv in self.kvs)
        net = np.asanyarray(coeffs)
        if net.ndim == 1:                                   # flat vector of scalar coefficients
            assert net.size == int(np.prod(grid)), 'Wrong length of coefficient vector'
            net = net.reshape(grid)
        assert net.shape[:self.sdim] == grid, 'Wrong shape of coefficients'
        tail = net.shape[self.sdim:]                        # () scalar-valued, (d,) vector-valued
        assert len(tail) <= 1, 'Tensor-valued NURBS functions not implemented'
        if weights is None:
            # the weight is the last component of `coeffs`
            assert tail and tail[0] > 1, 'Weights must be specified in the coeffs array'
            homog = net
            self._isscalar = False
            self.dim = tail[0] - 1
        else:
            w = np.asanyarray(weights)
            assert w.shape == grid, 'Wrong shape of weights array'
            self._isscalar = not tail
            self.dim = tail[0] if tail else 1
            comps = net[..., None] if not tail else net
            homog = np.concatenate((comps, w[..., None]), axis=-1)
        if not premultiplied:
            homog[..., :-1] *= homog[..., -1:]
        self.coeffs = homog

    def output_shape(self):
        return () if self._isscalar else (self.dim,)

    def copy(self):
        return NurbsFunc(tuple(kv.copy() for kv in self.kvs), self.coeffs.copy(), None, premultiplied=True)

    def coeffs_weights(self):
        """Non-premultiplied coefficients and weights."""
        w = self.coeffs[..., -1]
        return self.coeffs[..., :-1] / w[..., None], w.copy()

    def grid_eval(self, gridaxes):
        assert len(gridaxes) == self.sdim, 'Input has wrong dimension'
        f = _device_grid_eval(self.kvs, self.coeffs, True, self.dim, gridaxes, want_jac=False)
        return np.squeeze(f, -1) if self._isscalar else f

    def grid_jacobian(self, gridaxes):
        """Quotient rule on the homogeneous spline (pyiga/geometry.py:17-25,116-123), on the device."""
        assert len(gridaxes) == self.sdim, 'Input has wrong dimension'
        J = _device_grid_eval(self.kvs, self.coeffs, True, self.dim, gridaxes, want_jac=True)
        return np.squeeze(J, -2) if self._isscalar else J

    def parametric_derivatives(self, gridaxes):
        """(value, first, second) parametric derivatives of the rational function on a tensor grid: arrays of shape
        ``shape(grid) x dim``, ``... x dim x sdim`` and ``... x dim x sdim x sdim`` (derivative indices in (x, y, z) order, x =
        last grid axis), by the quotient rule on the homogeneous spline N / w:
            d_a f = (d_a N - f d_a w) / w,   d_ab f = (d_ab N - d_a f d_b w - d_b f d_a w - f d_ab w) / w."""
        sd = self.sdim
        part = bspline.tensor_partials(self.kvs, self.coeffs, gridaxes, 2)
        o = lambda *xyz: part[bspline._orders_of(sd, *xyz)]
        N, w = o()[..., :-1], o()[..., -1:]
        f = N / w
        d1 = np.stack([(o(a)[..., :-1] - f * o(a)[..., -1:]) / w for a in range(sd)], axis=-1)
        d2 = np.empty(f.shape + (sd, sd))
        for a in range(sd):
            for b in range(a, sd):
                h = (o(a, b)[..., :-1] - d1[..., a] * o(b)[..., -1:] - d1[..., b] * o(a)[..., -1:] - f * o(a, b)[..., -1:]) / w
                d2[..., a, b] = h
                d2[..., b, a] = h
        return f, d1, d2

    def grid_hessian(self, gridaxes):
        """Second derivatives, symmetric part linearised as (xx, xy, yy) resp. (xx, xy, xz, yy, yz, zz); shape
        ``shape(grid) x dim x num_hess``, the `dim` axis dropped for scalar functions (pyiga/geometry.py:125-150)."""
        assert len(gridaxes) == self.sdim, 'Input has wrong dimension'
        d2 = self.parametric_derivatives(gridaxes)[2]
        H = np.stack([d2[..., i, j] for i, j in bspline._hessian_index_pairs(self.sdim)], axis=-1)
        return np.squeeze(H, -2) if self._isscalar else H

    def boundary(self, bdspec):
        """One face of the parameter domain as a NURBS function with `sdim` reduced by one (pyiga/geometry.py:188-210): the
        first / last layer of the homogeneous net along that axis."""
        from .form_assemblers import parse_bdspec
        axis, side = parse_bdspec(bdspec, self.sdim)
        layer = np.take(self.coeffs, 0 if side == 0 else -1, axis=axis)
        return NurbsFunc(self.kvs[:axis] + self.kvs[axis + 1:], layer.copy(), None, premultiplied=True)

    def as_nurbs(self):
        return self

    def as_vector(self):
        if self.is_vector():
            return self
        assert self.is_scalar()
        vals, w = self.coeffs_weights()
        return NurbsFunc(self.kvs, vals, w)


def split_axis0(geo, tol=8.0):
    """Is the 3D spline map `geo` SEPARABLE along its first parametric axis -- G(xi0, xi1, xi2) = (g(xi1, xi2), z(xi0)) in
    some order of the components, an extruded cross-section such as ``tensor_product(line_segment(..), quarter_annulus())``?
    Decided on the control net alone (up to `tol` units in the last place, what forming the products of a tensor-product net
    costs).  Returns ``(zc, geo2d)`` -- the control values of the polynomial z on ``geo.kvs[0]`` and the 2D map of the
    cross-section -- or None.  (A rational z, i.e. weights that vary along axis 0, is not recognised.)"""
    if not isinstance(geo, (BSplineFunc, NurbsFunc)) or geo.sdim != 3 or geo.dim != 3:
        return None
    C = np.asarray(geo.coeffs, dtype=float)
    nurbs = isinstance(geo, NurbsFunc)
    eps = tol * np.finfo(float).eps
    same = lambda a, b: np.abs(a - b).max() <= eps * max(np.abs(a).max(), np.abs(b).max(), 1e-300)
    W = C[..., 3] if nurbs else None
    if nurbs and not all(same(W[a], W[0]) for a in range(1, W.shape[0])):
        return None                                         # weights vary along axis 0
    for cz in range(3):
        xy = [c for c in range(3) if c != cz]
        if not all(same(C[a][..., xy], C[0][..., xy]) for a in range(1, C.shape[0])):
            continue                                        # the other two components must not depend on axis 0
        Z = C[..., cz]
        ref = W[0] if nurbs else np.ones_like(Z[0])         # homogeneous: Z = z(a0) * w(a1, a2)
        k = np.unravel_index(np.argmax(np.abs(ref)), ref.shape)
        zc = Z[(slice(None),) + k] / ref[k]
        if not same(Z, zc[:, None, None] * ref[None]):
            continue
        if np.ptp(zc) == 0.0:
            return None                                     # degenerate: no extent along axis 0
        if nurbs:
            g2 = NurbsFunc(geo.kvs[1:], np.concatenate((C[0][..., xy], W[0][..., None]), axis=-1).copy(), None, premultiplied=True)
        else:
            g2 = BSplineFunc(geo.kvs[1:], C[0][..., xy].copy())
        return np.array(zc, dtype=float), g2
    return None


# ---------------------------------------------------------------------------------------------
# 2D geometries
def unit_square(num_intervals=1):
    """Unit square (pyiga/geometry.py:425-431)."""
    return unit_cube(dim=2, num_intervals=num_intervals)


def bspline_quarter_annulus(r1=1.0, r2=2.0):
    """B-spline approximation of a quarter annulus (pyiga/geometry.py:445-466)."""
    kvx = bspline.make_knots(1, 0.0, 1.0, 1)
    kvy = bspline.make_knots(2, 0.0, 1.0, 1)
    coeffs = np.array([
        [[r1, 0.0], [r2, 0.0]],
        [[r1, r1], [r2, r2]],
        [[0.0, r1], [0.0, r2]],
    ])
    return BSplineFunc((kvy, kvx), coeffs)


def quarter_annulus(r1=1.0, r2=2.0):
    """Exact NURBS quarter annulus in the first quadrant (pyiga/geometry.py:468-491)."""
    kvx = bspline.make_knots(1, 0.0, 1.0, 1)
    kvy = bspline.make_knots(2, 0.0, 1.0, 1)
    w = 1.0 / np.sqrt(2.0)
    coeffs = np.array([
        [[r1, 0.0, 1.0], [r2, 0.0, 1.0]],
        [[r1, r1, w], [r2, r2, w]],
        [[0.0, r1, 1.0], [0.0, r2, 1.0]],
    ])
    return NurbsFunc((kvy, kvx), coeffs, weights=None)


# ---------------------------------------------------------------------------------------------
# 3D geometries
def unit_cube(dim=3, num_intervals=1):
    """`dim`-dimensional unit cube (pyiga/geometry.py:533-540)."""
    return functools.reduce(tensor_product, dim * (line_segment(0.0, 1.0, intervals=num_intervals),))


def twisted_box():
    """Box with a twisted, bent right face (pyiga/geometry.py:557-589; G+Smo's
    twistedFlatQuarterAnnulus)."""
    kv1 = bspline.make_knots(1, 0.0, 1.0, 1)
    kv2 = bspline.make_knots(3, 0.0, 1.0, 1)
    coeffs = np.array([
        1, 0, 0, 2, 0, 0, 1, 0.5, 0, 2, 1.5, 0,
        0.5, 1, 0.5, 1.5, 2, 0.5, 0, 1, 2, 0, 2, 2,
        1, 0, 1, 2, 0, 1, 1, 0.5, 1, 2, 1.5, 1,
        1, 1, 1.5, 1.5, 2, 1.5, 1, 1, 2, 1, 2, 2,
    ], dtype=float).reshape((2, 4, 2, 3))
    return BSplineFunc((kv1, kv2, kv1), coeffs)


# ---------------------------------------------------------------------------------------------
# curves and products
def line_segment(x0, x1, support=(0.0, 1.0), intervals=1):
    """Straight line from `x0` to `x1` as a linear spline (pyiga/geometry.py:595-615)."""
    if np.isscalar(x0):
        x0 = [x0]
    if np.isscalar(x1):
        x1 = [x1]
    assert len(x0) == len(x1), 'Vectors must have same dimension'
    x0 = np.array(x0, dtype=float).ravel()
    x1 = np.array(x1, dtype=float).ravel()
    S = np.linspace(0.0, 1.0, intervals + 1).reshape((intervals + 1, 1))
    coeffs = (1 - S) * x0 + S * x1
    return BSplineFunc(bspline.make_knots(1, support[0], support[1], intervals), coeffs)


def _split_control_net(G):
    """(plain coefficients, weights or None) with a trailing component axis."""
    if isinstance(G, NurbsFunc):
        return G.coeffs_weights()
    C = G.coeffs if G.is_vector() else G.coeffs[..., None]
    return C, None


def tensor_product(G1, G2, *Gs):
    """``G(x,y) = G2(x) x G1(y)``: components are joined (x first), parameter axes are
    concatenated (y first) -- pyiga/geometry.py:755-809.  NURBS if any factor is NURBS."""
    if Gs:
        return tensor_product(G1, tensor_product(G2, *Gs))
    for G in (G1, G2):
        assert G.is_scalar() or G.is_vector(), 'only implemented for scalar- or vector-valued functions'
    C1, W1 = _split_control_net(G1)
    C2, W2 = _split_control_net(G2)
    n1, n2 = C1.shape[:G1.sdim], C2.shape[:G2.sdim]
    ones1, ones2 = (1,) * G1.sdim, (1,) * G2.sdim
    # broadcast both nets over the joint index space (axes of G1 first)
    B1 = np.broadcast_to(C1.reshape(n1 + ones2 + C1.shape[G1.sdim:]), n1 + n2 + C1.shape[G1.sdim:])
    B2 = np.broadcast_to(C2.reshape(ones1 + n2 + C2.shape[G2.sdim:]), n1 + n2 + C2.shape[G2.sdim:])
    C = np.concatenate((B2, B1), axis=-1)         # coefficients in (x, y) order
    kvs = tuple(G1.kvs) + tuple(G2.kvs)
    if W1 is None and W2 is None:
        return BSplineFunc(kvs, C)
    W1 = np.ones(n1) if W1 is None else W1
    W2 = np.ones(n2) if W2 is None else W2
    W = W1.reshape(n1 + ones2) * W2.reshape(ones1 + n2)
    return NurbsFunc(kvs, C, W)
