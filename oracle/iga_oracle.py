"""CPU oracle for the tensor-product mass/stiffness assembly hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain numpy/C restatement of the
reference algorithm (c-f-h/pyiga, snapshot 2025-02-22) for the path
`assemble.stiffness()/mass()` with a geometry map.  It is imported only by
`tests/`, by `__graft_entry__.smoke()` and by the `cpu_baseline` leg of
`bench.py` -- never by the product package `pyiga_amd`, which has no CPU
fallback.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function
below against
  * the reference's own fixtures  test/poisson_neu_d{2,3}_*.mtx.gz
    (committed unchanged under tests/golden/), tolerance 1e-14 absolute as in
    test/test_assemble.py:138-168,
  * the literal 1D matrices of test/test_assemble.py:10-40,
  * golden vectors produced by running the real reference in the build
    container (tests/golden/make_golden.py, outputs tests/golden/golden_*.npz).

Every function cites the reference file:line it restates (paths relative to the
reference checkout).  Summation orders follow the reference: entries are
computed entry-by-entry over the intersection of supports
(pyiga/assemblers.pyx:1455-1540), not element-by-element.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import scipy.sparse

_HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------
# knot vectors                                          pyiga/bspline.py:36-213
class KnotVector:
    """Open knot vector + degree (pyiga/bspline.py:36-144)."""

    def __init__(self, knots, p):
        self.kv = np.ascontiguousarray(knots, dtype=np.float64)
        assert np.all(self.kv[1:] - self.kv[:-1] >= 0), 'knots should be increasing'
        self.p = int(p)
        # pyiga/bspline.py:110-113: unique knots + inverse map
        self.mesh, self.knots_to_mesh = np.unique(self.kv, return_inverse=True)

    @property
    def numdofs(self):                       # pyiga/bspline.py:82-85
        return self.kv.size - self.p - 1

    @property
    def numspans(self):                      # pyiga/bspline.py:87-90
        return self.mesh.size - 1

    def mesh_support_idx_all(self):          # pyiga/bspline.py:129-136
        n = self.numdofs
        startend = np.stack((np.arange(0, n), np.arange(self.p + 1, n + self.p + 1)), axis=1)
        return self.knots_to_mesh[startend]

    def mesh_span_indices(self):             # pyiga/bspline.py:138-144
        k2m = self.knots_to_mesh
        return np.where(k2m[1:] != k2m[:-1])[0]


def make_knots(p, a, b, n, mult=1):
    """pyiga/bspline.py:192-213 (same np.arange expression => same bits)."""
    kv = np.concatenate(
        (np.repeat(a, p + 1),
         np.repeat(np.arange(a, b, (b - a) / n)[1:], mult),
         np.repeat(b, p + 1)))
    return KnotVector(kv, p)


# ---------------------------------------------------------------------------
# quadrature                                           pyiga/quadrature.py:3-23
def gauss_rule(deg, a, b):
    m = 0.5 * (a + b)
    h = 0.5 * (b - a)
    x, w = np.polynomial.legendre.leggauss(deg)
    nodes = (np.outer(h, x) + m[:, np.newaxis])
    weights = np.outer(h, w)
    return nodes.ravel(), weights.ravel()


def make_tensor_quadrature(meshes, nqp):
    gauss = tuple(gauss_rule(nqp, mesh[:-1], mesh[1:]) for mesh in meshes)
    return tuple(g[0] for g in gauss), tuple(g[1] for g in gauss)


# ---------------------------------------------------------------------------
# B-spline evaluation                                pyiga/bspline_cy.pyx:13-145
def findspan(kv, p, u):
    """Binary search, pyiga/bspline_cy.pyx:13-27."""
    n = kv.shape[0]
    if u >= kv[n - p - 1]:
        return n - p - 2
    a, b = 0, n - 1
    while b - a > 1:
        c = a + (b - a) // 2
        if kv[c] > u:
            b = c
        else:
            a = c
    return a


def active_deriv_single(kv, p, u, numderiv):
    """Piegl-Tiller A2.3 as in pyiga/bspline_cy.pyx:42-121."""
    NDU = np.empty((p + 1, p + 1))
    result = np.empty((numderiv + 1, p + 1))
    left = np.empty(p + 1)
    right = np.empty(p + 1)
    span = findspan(kv, p, u)
    NDU[0, 0] = 1.0
    for j in range(1, p + 1):
        left[j - 1] = u - kv[span + 1 - j]
        right[j - 1] = kv[span + j] - u
        saved = 0.0
        for r in range(j):
            NDU[j, r] = right[r] + left[j - r - 1]
            temp = NDU[r, j - 1] / NDU[j, r]
            NDU[r, j] = saved + right[r] * temp
            saved = left[j - r - 1] * temp
        NDU[j, j] = saved
    for j in range(p + 1):
        result[0, j] = NDU[j, p]
    a1 = np.empty(p + 2)
    a2 = np.empty(p + 2)
    for r in range(p + 1):
        a1[0] = 1.0
        fac = p
        for k in range(1, numderiv + 1):
            rk = r - k
            pk = p - k
            d = 0.0
            if r >= k:
                a2[0] = a1[0] / NDU[pk + 1, rk]
                d = a2[0] * NDU[rk, pk]
            j1 = 1 if rk >= -1 else -rk
            j2 = k - 1 if r - 1 <= pk else p - r
            for j in range(j1, j2 + 1):
                a2[j] = (a1[j] - a1[j - 1]) / NDU[pk + 1, rk + j]
                d += a2[j] * NDU[rk + j, pk]
            if r <= pk:
                a2[k] = -a1[k - 1] / NDU[pk + 1, r]
                d += a2[k] * NDU[r, pk]
            result[k, r] = d * fac
            fac *= pk
            a1, a2 = a2, a1
    return result


def active_deriv(knotvec, u, numderiv):
    """(numderiv+1, p+1, len(u)) -- pyiga/bspline_cy.pyx:126-145."""
    u = np.asarray(u, dtype=np.float64)
    out = np.empty((numderiv + 1, knotvec.p + 1, u.shape[0]))
    for i in range(u.shape[0]):
        out[:, :, i] = active_deriv_single(knotvec.kv, knotvec.p, float(u[i]), numderiv)
    return out


def first_active(knotvec, u):
    """pyx_findspans(...) - p, pyiga/bspline.py:636,658."""
    return np.array([findspan(knotvec.kv, knotvec.p, float(x)) for x in u], dtype=np.int64) - knotvec.p


def collocation_derivs_dense(knotvec, nodes, derivs):
    """Dense version of pyiga/bspline.py:629-646: list of (len(nodes), numdofs)."""
    vals = active_deriv(knotvec, nodes, derivs)       # (derivs+1, p+1, m)
    fa = first_active(knotvec, nodes)
    m, n, p = len(nodes), knotvec.numdofs, knotvec.p
    mats = [np.zeros((m, n)) for _ in range(derivs + 1)]
    for d in range(derivs + 1):
        for k in range(m):
            mats[d][k, fa[k]:fa[k] + p + 1] = vals[d, :, k]
    return mats


def compute_values_derivs(knotvec, grid, derivs):
    """Axes (basis function, grid point, derivative): pyiga/assemble_tools.py:7-12."""
    colloc = collocation_derivs_dense(knotvec, grid, derivs)
    return np.ascontiguousarray(np.stack([X.T for X in colloc], axis=-1))


# ---------------------------------------------------------------------------
# geometry                        pyiga/bspline.py:874-921, geometry.py:17-123
def apply_tprod_dense(ops, A):
    """pyiga/tensor.py:97-128 for dense operators."""
    n = len(ops)
    for i in reversed(range(n)):
        A = np.tensordot(ops[i], A, axes=([1], [n - 1]))
    return A


def bspline_grid_eval(kvs, coeffs, gridaxes):
    colloc = [collocation_derivs_dense(kv, g, 0)[0] for kv, g in zip(kvs, gridaxes)]
    return apply_tprod_dense(colloc, coeffs)


def bspline_grid_jacobian(kvs, coeffs, gridaxes):
    """pyiga/bspline.py:897-921; last axis = d/d(x,y,z), x = LAST grid axis."""
    sdim = len(kvs)
    colloc = [collocation_derivs_dense(kv, g, 1) for kv, g in zip(kvs, gridaxes)]
    comps = []
    for i in reversed(range(sdim)):
        ops = [colloc[j][1 if j == i else 0] for j in range(sdim)]
        comps.append(apply_tprod_dense(ops, coeffs))
    return np.stack(comps, axis=-1)


def nurbs_grid_jacobian(kvs, coeffs, gridaxes):
    """pyiga/geometry.py:17-25,116-123 (coeffs premultiplied, weight last)."""
    val = bspline_grid_eval(kvs, coeffs, gridaxes)
    jac = bspline_grid_jacobian(kvs, coeffs, gridaxes)
    V = val[..., :-1, None]
    W = val[..., -1:, None]
    Vjac = jac[..., :-1, :]
    Wjac = jac[..., -1:, :]
    return (Vjac * W - V * Wjac) / (W ** 2)


def grid_eval(geo, gridaxes):
    """BSplineFunc/NurbsFunc.grid_eval: pyiga/bspline.py:874-895, pyiga/geometry.py:103-114."""
    vals = bspline_grid_eval(geo['kvs'], geo['coeffs'], gridaxes)
    if geo['nurbs']:
        return vals[..., :-1] / vals[..., -1:]
    return vals


def grid_jacobian(geo, gridaxes):
    """geo = dict(kvs=[KnotVector...], coeffs=ndarray, nurbs=bool)."""
    if geo['nurbs']:
        return nurbs_grid_jacobian(geo['kvs'], geo['coeffs'], gridaxes)
    return bspline_grid_jacobian(geo['kvs'], geo['coeffs'], gridaxes)


# control nets of the geometries the configs use (pyiga/geometry.py:425-615)
def geo_bspline_quarter_annulus(r1=1.0, r2=2.0):         # geometry.py:445-466
    kvx = make_knots(1, 0.0, 1.0, 1)
    kvy = make_knots(2, 0.0, 1.0, 1)
    coeffs = np.array([[[r1, 0.0], [r2, 0.0]], [[r1, r1], [r2, r2]], [[0.0, r1], [0.0, r2]]])
    return dict(kvs=[kvy, kvx], coeffs=coeffs, nurbs=False)


def geo_quarter_annulus(r1=1.0, r2=2.0):                 # geometry.py:468-491,88-90
    kvx = make_knots(1, 0.0, 1.0, 1)
    kvy = make_knots(2, 0.0, 1.0, 1)
    s = 1.0 / np.sqrt(2.0)
    coeffs = np.array([[[r1, 0.0, 1.0], [r2, 0.0, 1.0]],
                       [[r1, r1, s], [r2, r2, s]],
                       [[0.0, r1, 1.0], [0.0, r2, 1.0]]])
    coeffs[..., :-1] *= coeffs[..., -1:]                  # premultiply by weights
    return dict(kvs=[kvy, kvx], coeffs=coeffs, nurbs=True)


def geo_twisted_box():                                    # geometry.py:557-589
    kv1 = make_knots(1, 0.0, 1.0, 1)
    kv2 = make_knots(3, 0.0, 1.0, 1)
    c = np.array([1, 0, 0, 2, 0, 0, 1, 0.5, 0, 2, 1.5, 0, 0.5, 1, 0.5, 1.5, 2, 0.5, 0, 1, 2, 0, 2, 2,
                  1, 0, 1, 2, 0, 1, 1, 0.5, 1, 2, 1.5, 1, 1, 1, 1.5, 1.5, 2, 1.5, 1, 1, 2, 1, 2, 2],
                 dtype=float).reshape((2, 4, 2, 3))
    return dict(kvs=[kv1, kv2, kv1], coeffs=c, nurbs=False)


def geo_unit_cube(dim):                                   # geometry.py:533-540
    kv = make_knots(1, 0.0, 1.0, 1)
    shape = (2,) * dim + (dim,)
    c = np.zeros(shape)
    for ax in range(dim):
        idx = [None] * dim
        idx[ax] = slice(None)
        # component ordering: coefficients are in (x, y, z) order, axes in (z, y, x)
        c[..., dim - 1 - ax] += np.array([0.0, 1.0])[tuple(idx)]
    return dict(kvs=[kv] * dim, coeffs=c, nurbs=False)


def geo_cylinder(r1=1.0, r2=2.0):
    """tensor_product(line_segment(0,1), quarter_annulus()) -- geometry.py:755-809."""
    ann = geo_quarter_annulus(r1, r2)
    ca = ann['coeffs']                                    # (3,2,3) premultiplied (x*w, y*w, w)
    wts = ca[..., -1]
    xy = ca[..., :-1] / wts[..., None]                    # coeffs_weights(): un-premultiplied
    z = np.array([0.0, 1.0])
    C = np.empty((2, 3, 2, 4))
    C[..., 0:2] = xy[None]                                # (x, y) from the annulus (G2)
    C[..., 2] = z[:, None, None]                          # z from the line segment (G1)
    C[..., 3] = wts[None]
    C[..., :-1] *= C[..., -1:]                            # NurbsFunc.__init__ premultiplies
    kvz = make_knots(1, 0.0, 1.0, 1)
    return dict(kvs=[kvz] + ann['kvs'], coeffs=C, nurbs=True)


# ---------------------------------------------------------------------------
# fields                                  pyiga/assemblers.pyx:86-110,234-275,
#                                                       1223-1249,1389-1449
def precompute_fields(kind, jac, gw):
    """kind in {'mass','stiffness'}; jac (N..., d, d); gw tuple of weight vectors.

    Returns fields (N..., F) with F=1 (W) or d(d+1)/2 (upper triangle of
    B = W * Jinv Jinv^T, row-major).  Arithmetic follows the generated code.
    """
    d = jac.shape[-1]
    t = jac.reshape(jac.shape[:-2] + (d * d,))
    if d == 2:
        GW = gw[0][:, None] * gw[1][None, :]
        det = t[..., 0] * t[..., 3] - t[..., 1] * t[..., 2]
        if kind == 'mass':
            return (GW * np.abs(det))[..., None]
        W = GW * np.abs(det)
        i = 1.0 / det
        J0, J1, J2, J3 = i * t[..., 3], i * -t[..., 1], i * -t[..., 2], i * t[..., 0]
        return np.stack((W * (J0 * J0 + J1 * J1), W * (J0 * J2 + J1 * J3), W * (J2 * J2 + J3 * J3)), axis=-1)
    GW = (gw[0][:, None, None] * gw[1][None, :, None]) * gw[2][None, None, :]
    t3 = t[..., 4] * t[..., 8] - t[..., 5] * t[..., 7]
    t4 = t[..., 3] * t[..., 8] - t[..., 5] * t[..., 6]
    t5 = t[..., 3] * t[..., 7] - t[..., 4] * t[..., 6]
    det = (t[..., 0] * t3 - t[..., 1] * t4) + t[..., 2] * t5
    W = GW * np.abs(det)
    if kind == 'mass':
        return W[..., None]
    i = 1.0 / det
    JI = [i * t3,
          i * -(t[..., 1] * t[..., 8] - t[..., 2] * t[..., 7]),
          i * (t[..., 1] * t[..., 5] - t[..., 2] * t[..., 4]),
          i * -t4,
          i * (t[..., 0] * t[..., 8] - t[..., 2] * t[..., 6]),
          i * -(t[..., 0] * t[..., 5] - t[..., 2] * t[..., 3]),
          i * t5,
          i * -(t[..., 0] * t[..., 7] - t[..., 1] * t[..., 6]),
          i * (t[..., 0] * t[..., 4] - t[..., 1] * t[..., 3])]
    B = [W * ((JI[0] * JI[0] + JI[1] * JI[1]) + JI[2] * JI[2]),
         W * ((JI[0] * JI[3] + JI[1] * JI[4]) + JI[2] * JI[5]),
         W * ((JI[0] * JI[6] + JI[1] * JI[7]) + JI[2] * JI[8]),
         W * ((JI[3] * JI[3] + JI[4] * JI[4]) + JI[5] * JI[5]),
         W * ((JI[3] * JI[6] + JI[4] * JI[7]) + JI[5] * JI[8]),
         W * ((JI[6] * JI[6] + JI[7] * JI[7]) + JI[8] * JI[8])]
    return np.stack(B, axis=-1)


def precompute_fields_convdiff(jac, xphys, coeff, gw):
    """Fields of the generated assembler for the convection-diffusion form (3D): per point
    [diff_coeff, x, y, z, W, JacInv(9 row-major)] -- layout and arithmetic of the code the reference
    generates at run time (pyiga/codegen/cython.py:673-701; same cofactor formulas as
    assemblers.pyx:1420-1441)."""
    t = jac.reshape(jac.shape[:-2] + (9,))
    GW = (gw[0][:, None, None] * gw[1][None, :, None]) * gw[2][None, None, :]
    t6 = t[..., 4] * t[..., 8] - t[..., 5] * t[..., 7]
    t7 = t[..., 3] * t[..., 8] - t[..., 5] * t[..., 6]
    t8 = t[..., 3] * t[..., 7] - t[..., 4] * t[..., 6]
    det = (t[..., 0] * t6 - t[..., 1] * t7) + t[..., 2] * t8
    i = 1.0 / det
    JI = [i * t6,
          i * -(t[..., 1] * t[..., 8] - t[..., 2] * t[..., 7]),
          i * (t[..., 1] * t[..., 5] - t[..., 2] * t[..., 4]),
          i * -t7,
          i * (t[..., 0] * t[..., 8] - t[..., 2] * t[..., 6]),
          i * -(t[..., 0] * t[..., 5] - t[..., 2] * t[..., 3]),
          i * t8,
          i * -(t[..., 0] * t[..., 7] - t[..., 1] * t[..., 6]),
          i * (t[..., 0] * t[..., 4] - t[..., 1] * t[..., 3])]
    return np.stack([coeff, xphys[..., 0], xphys[..., 1], xphys[..., 2], GW * np.abs(det)] + JI, axis=-1)


# ---------------------------------------------------------------------------
# sparsity                 pyiga/mlmatrix.py:420-440, mlmatrix_cy.pyx:189-289
def precompute_fields_form(jac, table, gw):
    """Fields of a general first-order-jet form: [P_rs (16, zeros where absent), W, JacInv (9)] per Gauss
    point -- what the reference's generated precompute step stores for such a vform: the inputs, W and
    JacInv (pyiga/vform.py:229-239, pyiga/codegen/cython.py:673-701)."""
    G = jac.shape[:-2]
    det = np.linalg.det(jac)
    W = np.abs(det)
    for k, w in enumerate(gw):
        W = W * w.reshape((1,) * k + (-1,) + (1,) * (len(gw) - 1 - k))
    JI = np.linalg.inv(jac)                    # [param (x,y,z order)][physical]: d xi_a / d x_r
    d = len(G)
    F = np.zeros(G + (17 + d * d,))
    for r in range(d + 1):
        for s in range(d + 1):
            if table[r][s] is not None:
                F[..., 4 * r + s] = np.broadcast_to(table[r][s], G)
    F[..., 16] = W
    F[..., 17:17 + d * d] = JI.reshape(G + (d * d,))
    return F


def compute_sparsity_ij(kv1, kv2):
    ms1 = kv1.mesh_support_idx_all()
    ms2 = kv2.mesh_support_idx_all()
    IJ = []
    for i in range(ms2.shape[0]):
        j = int(np.searchsorted(ms1[:, 1], ms2[i, 0], side='right'))
        while j < ms1.shape[0] and min(ms2[i, 1], ms1[j, 1]) > max(ms2[i, 0], ms1[j, 0]):
            IJ.append((i, j))
            j += 1
    return np.array(IJ, dtype=np.uint32)


def ml_nonzero(bidx, bs, lower_tri=False):
    """Kronecker expansion in the reference's (pair0, pair1, pair2) order."""
    L = len(bidx)
    I = bidx[0][:, 0].astype(np.int64)
    J = bidx[0][:, 1].astype(np.int64)
    for k in range(1, L):
        mk, nk = bs[k]
        I = (I[:, None] * mk + bidx[k][None, :, 0].astype(np.int64)).ravel()
        J = (J[:, None] * nk + bidx[k][None, :, 1].astype(np.int64)).ravel()
    if lower_tri:
        keep = J <= I
        I, J = I[keep], J[keep]
    return I, J


# ---------------------------------------------------------------------------
# C kernels (entry-wise combine loops)
_lib = None


def _cflags(fast):
    # `fast` mirrors the reference build (setup.py:10-18: -O3 -march=native
    # -ffast-math) but with a portable ISA level, because the .so is built in
    # one container and may run on a different host CPU.
    if fast:
        return ['-O3', '-ffast-math', '-march=x86-64-v3']
    return ['-O2', '-ffp-contract=off']


def build(force=False):
    """Compile oracle/oracle_kernels.c -> oracle/_build/liboracle{,_fast}.so."""
    src = os.path.join(_HERE, 'oracle_kernels.c')
    bdir = os.path.join(_HERE, '_build')
    os.makedirs(bdir, exist_ok=True)
    outs = []
    for fast in (False, True):
        out = os.path.join(bdir, 'liboracle_fast.so' if fast else 'liboracle.so')
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            cmd = ['gcc', '-shared', '-fPIC', '-fopenmp'] + _cflags(fast) + ['-o', out, src, '-lm']
            subprocess.check_call(cmd)
        outs.append(out)
    return outs


def _load(fast=False):
    global _lib
    if _lib is None:
        _lib = {}
    if fast not in _lib:
        strict, fst = build()
        lib = ctypes.CDLL(fst if fast else strict)
        lib.orc_entries.restype = ctypes.c_int
        lib.orc_entries.argtypes = [
            ctypes.c_int, ctypes.c_int,                      # dim, kind
            ctypes.c_void_p,                                 # ndofs[dim] (size_t)
            ctypes.c_void_p,                                 # ngauss[dim] (size_t)
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,   # meshsupp0..2 (int64, N x 2, scaled by nqp)
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,   # C0..2 (N x ng x nder)
            ctypes.c_int,                                    # nder
            ctypes.c_void_p, ctypes.c_int,                   # fields, F
            ctypes.c_void_p, ctypes.c_size_t,                # idx (M x 2 size_t), M
            ctypes.c_void_p, ctypes.c_int]                   # out, nthreads
        _lib[fast] = lib
    return _lib[fast]


class Assembler:
    """Restates *Assembler{2,3}D.__init__ (pyiga/assemblers.pyx:38-80,186-228,
    1170-1217,1336-1383) + entry/multi_entries (pyiga/genericasm.pxi:677-758)."""

    def __init__(self, kind, kvs, geo=None, jac=None, coeff=None, table=None):
        assert kind in ('mass', 'stiffness', 'convdiff', 'form')
        self.kind = kind
        self.kvs = tuple(kvs)
        self.dim = len(kvs)
        assert self.dim in (2, 3)
        self.nqp = max(kv.p for kv in kvs) + 1
        self.grid, self.gw = make_tensor_quadrature([kv.mesh for kv in kvs], self.nqp)
        self.nder = 1 if kind == 'mass' else 2
        self.meshsupp = [np.ascontiguousarray(self.nqp * kv.mesh_support_idx_all(), dtype=np.int64)
                         for kv in kvs]
        self.C = [compute_values_derivs(kv, g, self.nder - 1) for kv, g in zip(kvs, self.grid)]
        if jac is None:
            jac = grid_jacobian(geo, self.grid)
        self.jac = np.ascontiguousarray(jac)
        if kind == 'convdiff':
            # grid_eval_transformed(diff_coeff, grid, geo): pyiga/utils.py:43-52
            assert self.dim == 3
            xphys = grid_eval(geo, self.grid)
            c = coeff(xphys[..., 0], xphys[..., 1], xphys[..., 2]) * np.ones(xphys.shape[:-1])
            self.fields = np.ascontiguousarray(precompute_fields_convdiff(self.jac, xphys, c, self.gw))
        elif kind == 'form':
            # `table`: 4x4 nested list of functions of the physical coordinates (or constants / None), P[r][s]
            xphys = grid_eval(geo, self.grid)
            G = xphys.shape[:-1]
            vals = [[None if e is None else np.broadcast_to(e(*(xphys[..., k] for k in range(self.dim))) if callable(e) else e, G)
                     for e in row] for row in table]
            self.fields = np.ascontiguousarray(precompute_fields_form(self.jac, vals, self.gw))
        else:
            self.fields = np.ascontiguousarray(precompute_fields(kind, self.jac, self.gw))
        self.ndofs = np.array([kv.numdofs for kv in kvs], dtype=np.uintp)
        self.ngauss = np.array([g.shape[0] for g in self.grid], dtype=np.uintp)

    def multi_entries(self, idx, nthreads=1, fast=False):
        idx = np.ascontiguousarray(idx, dtype=np.uintp)
        assert idx.ndim == 2 and idx.shape[1] == 2
        out = np.zeros(idx.shape[0])
        lib = _load(fast)
        ms = self.meshsupp + [None] * (3 - self.dim)
        C = self.C + [None] * (3 - self.dim)
        p = lambda a: None if a is None else a.ctypes.data
        rc = lib.orc_entries(self.dim, {'mass': 0, 'stiffness': 1, 'convdiff': 2, 'form': 3}[self.kind],
                             p(self.ndofs), p(self.ngauss),
                             p(ms[0]), p(ms[1]), p(ms[2]), p(C[0]), p(C[1]), p(C[2]),
                             self.nder, p(self.fields), self.fields.shape[-1],
                             p(idx), idx.shape[0], p(out), int(nthreads))
        assert rc == 0
        return out

    def entry(self, i, j):
        return float(self.multi_entries(np.array([[i, j]], dtype=np.uintp))[0])

    def entries_python(self, idx):
        """Pure-numpy version of the same sums (slow; cross-checks the C loops)."""
        out = np.zeros(len(idx))
        shape = tuple(int(n) for n in self.ndofs)
        for k, (I, J) in enumerate(idx):
            i = np.unravel_index(int(I), shape)
            j = np.unravel_index(int(J), shape)
            sl, vu, vv = [], [], []
            empty = False
            for ax in range(self.dim):
                a = max(self.meshsupp[ax][i[ax], 0], self.meshsupp[ax][j[ax], 0])
                b = min(self.meshsupp[ax][i[ax], 1], self.meshsupp[ax][j[ax], 1])
                if a >= b:
                    empty = True
                    break
                sl.append(slice(a, b))
                vu.append(self.C[ax][j[ax], a:b, :])
                vv.append(self.C[ax][i[ax], a:b, :])
            if empty:
                continue
            F = self.fields[tuple(sl)]
            d = self.dim
            if self.kind == 'mass':
                u = vu[0][:, 0]
                v = vv[0][:, 0]
                for ax in range(1, d):
                    u = np.multiply.outer(u, vu[ax][:, 0])
                    v = np.multiply.outer(v, vv[ax][:, 0])
                out[k] = np.sum(u * v * F[..., 0])
            else:
                def grad(vals):
                    comps = []
                    for c in range(d):             # component c differentiates axis d-1-c
                        t = None
                        for ax in range(d):
                            f = vals[ax][:, 1 if ax == d - 1 - c else 0]
                            t = f if t is None else np.multiply.outer(t, f)
                        comps.append(t)
                    return comps
                gu, gv = grad(vu), grad(vv)
                sym = {}
                n = 0
                for r in range(d):
                    for c in range(r, d):
                        sym[(r, c)] = sym[(c, r)] = n
                        n += 1
                tot = 0.0
                for r in range(d):
                    for c in range(d):
                        tot = tot + F[..., sym[(r, c)]] * gu[c] * gv[r]
                out[k] = np.sum(tot)
        return out


def local_entries(kind, kvs, geo, idx, coeff=None, table=None):
    """Entries (i, j) of a patch that is too large for Assembler (the fields of BASELINE configs 4 and 5 take 12.6 GB + the
    Jacobians): the same entry sums (orc_entries / genericasm.pxi:677-758) with everything geometry-dependent evaluated on
    the Gauss points of each pair's support intersection only -- what the reference's on-demand assemblers do with a bounding
    box (pyiga/codegen/cython.py:541-559).  Pairs with disjoint supports give 0.0."""
    kvs = tuple(kvs)
    dim = len(kvs)
    nqp = max(kv.p for kv in kvs) + 1
    shape = tuple(kv.numdofs for kv in kvs)
    supp = [kv.mesh_support_idx_all() for kv in kvs]
    out = np.zeros(len(idx))
    for k, (I, J) in enumerate(np.asarray(idx, dtype=np.int64)):
        i, j = np.unravel_index(int(I), shape), np.unravel_index(int(J), shape)
        cells = [(max(supp[a][i[a], 0], supp[a][j[a], 0]), min(supp[a][i[a], 1], supp[a][j[a], 1])) for a in range(dim)]
        if any(lo >= hi for lo, hi in cells):
            continue
        sub = Assembler.__new__(Assembler)
        sub.kind, sub.kvs, sub.dim, sub.nqp = kind, kvs, dim, nqp
        sub.nder = 1 if kind == 'mass' else 2
        sub.grid, sub.gw = make_tensor_quadrature([kv.mesh[lo:hi + 1] for kv, (lo, hi) in zip(kvs, cells)], nqp)
        ng = [g.shape[0] for g in sub.grid]
        # supports in local Gauss indices (clipped to the box; only the rows of i and j are read)
        sub.meshsupp = [np.ascontiguousarray(np.clip(nqp * (supp[a] - cells[a][0]), 0, ng[a]), dtype=np.int64) for a in range(dim)]
        sub.C = [compute_values_derivs(kv, g, sub.nder - 1) for kv, g in zip(kvs, sub.grid)]
        jac = np.ascontiguousarray(grid_jacobian(geo, sub.grid))
        if kind == 'convdiff':
            xphys = grid_eval(geo, sub.grid)
            c = coeff(xphys[..., 0], xphys[..., 1], xphys[..., 2]) * np.ones(xphys.shape[:-1])
            sub.fields = np.ascontiguousarray(precompute_fields_convdiff(jac, xphys, c, sub.gw))
        elif kind == 'form':
            xphys = grid_eval(geo, sub.grid)
            G = xphys.shape[:-1]
            vals = [[None if e is None else np.broadcast_to(e(*(xphys[..., c] for c in range(dim))) if callable(e) else e, G)
                     for e in row] for row in table]
            sub.fields = np.ascontiguousarray(precompute_fields_form(jac, vals, sub.gw))
        else:
            sub.fields = np.ascontiguousarray(precompute_fields(kind, jac, sub.gw))
        sub.ndofs = np.array(shape, dtype=np.uintp)
        sub.ngauss = np.array(ng, dtype=np.uintp)
        out[k] = sub.multi_entries(np.array([[I, J]], dtype=np.uintp))[0]
    return out


def lower_pattern(kvs):
    """MLStructure.from_kvs + nonzero(lower_tri=True): pyiga/mlmatrix.py:59-65,113-130."""
    bidx = [compute_sparsity_ij(kv, kv) for kv in kvs]
    bs = [(kv.numdofs, kv.numdofs) for kv in kvs]
    return ml_nonzero(bidx, bs, lower_tri=True)


def full_pattern(kvs):
    """MLStructure.from_kvs + nonzero(lower_tri=False)."""
    bidx = [compute_sparsity_ij(kv, kv) for kv in kvs]
    bs = [(kv.numdofs, kv.numdofs) for kv in kvs]
    return ml_nonzero(bidx, bs, lower_tri=False)


def assemble_nonsymmetric(kind, kvs, geo, coeff=None, nthreads=1, table=None):
    """assemble_entries(asm, symmetric=False): every pattern entry is computed directly."""
    asm = Assembler(kind, kvs, geo=geo, coeff=coeff, table=table)
    I, J = full_pattern(kvs)
    entries = asm.multi_entries(np.column_stack((I, J)), nthreads=nthreads)
    n = int(np.prod([kv.numdofs for kv in kvs]))
    return scipy.sparse.coo_matrix((entries, (I, J)), shape=(n, n)).tocsr()


def assemble(kind, kvs, geo=None, jac=None, nthreads=1, fast=False, return_timing=None):
    """assemble_entries(asm, symmetric=True): pyiga/assemble.py:703-754."""
    import time
    t0 = time.perf_counter()
    asm = Assembler(kind, kvs, geo=geo, jac=jac)
    t1 = time.perf_counter()
    I, J = lower_pattern(kvs)
    t2 = time.perf_counter()
    entries = asm.multi_entries(np.column_stack((I, J)), nthreads=nthreads, fast=fast)
    t3 = time.perf_counter()
    n = int(np.prod([kv.numdofs for kv in kvs]))
    A = scipy.sparse.coo_matrix((entries, (I, J)), shape=(n, n)).tocsr()
    off = np.nonzero(I != J)[0]
    A = A + scipy.sparse.coo_matrix((entries[off], (J[off], I[off])), shape=(n, n))
    A = A.tocsr()
    t4 = time.perf_counter()
    if return_timing is not None:
        return_timing.update(init=t1 - t0, nonzero=t2 - t1, entries=t3 - t2, csr=t4 - t3)
    return A


# ---------------------------------------------------------------------------
# 1D matrices + Kronecker path                   pyiga/assemble.py:125-190,236-282
def function_grid_eval(f, gridaxes, geo=None):
    """utils.grid_eval / grid_eval_transformed (pyiga/utils.py:33-52): `f` takes (x, y, z); tuples become
    a trailing component axis; results are broadcast to the full grid."""
    shape = tuple(len(g) for g in gridaxes)
    if geo is None:
        mesh = list(np.meshgrid(*gridaxes, sparse=True, indexing='ij'))
        mesh.reverse()
        vals = f(*mesh)
    else:
        X = grid_eval(geo, gridaxes)
        vals = f(*(X[..., i] for i in range(X.shape[-1])))
    if isinstance(vals, tuple):
        vals = np.stack([np.broadcast_to(np.asanyarray(v), shape) for v in vals], axis=-1)
    vals = np.asanyarray(vals)
    return np.array(np.broadcast_to(vals, shape + vals.shape[len(shape):]), dtype=float)


def inner_products(kvs, f, f_physical=False, geo=None):
    """Load vector, restating pyiga/assemble.py:288-340 step by step: function values on the Gauss grid,
    times the tensor quadrature weights, times |det J| when a geometry is given, then the transposed
    collocation matrices along every axis."""
    kvs = tuple(kvs)
    nqp = max(kv.p for kv in kvs) + 1
    grid, gw = make_tensor_quadrature([kv.mesh for kv in kvs], nqp)
    if f_physical:
        assert geo is not None, 'inner_products in physical domain requires geometry'
        fvals = function_grid_eval(f, grid, geo)
    else:
        fvals = function_grid_eval(f, grid)
    extra = fvals.ndim - len(kvs)
    for k, w in enumerate(gw):
        fvals = fvals * w.reshape((1,) * k + (-1,) + (1,) * (len(kvs) - 1 - k + extra))
    if geo is not None:
        det = np.abs(np.linalg.det(grid_jacobian(geo, grid)))
        fvals = fvals * det.reshape(det.shape + (1,) * extra)
    Ct = [collocation_derivs_dense(kv, g, 0)[0].T for kv, g in zip(kvs, grid)]
    return apply_tprod_dense(Ct, fvals)


def load_vector_jet(kvs, geo, jet):
    """Load vector of a functional in the first-order jet of v,  sum_r F_r D_r v  (D_0 = id, D_1.. physical
    derivatives), the way the reference's generated arity-1 assembler evaluates it: physical gradient of the
    basis function through JacInv, times the coefficients, times W, summed over the Gauss grid.
    jet: list of d+1 functions of the physical coordinates (or constants / None)."""
    kvs = tuple(kvs)
    d = len(kvs)
    nqp = max(kv.p for kv in kvs) + 1
    grid, gw = make_tensor_quadrature([kv.mesh for kv in kvs], nqp)
    X = grid_eval(geo, grid)
    G = X.shape[:-1]
    F = [np.zeros(G) if e is None else np.broadcast_to(e(*(X[..., k] for k in range(d))) if callable(e) else e, G) for e in jet]
    jac = grid_jacobian(geo, grid)
    W = np.abs(np.linalg.det(jac))
    for k, w in enumerate(gw):
        W = W * w.reshape((1,) * k + (-1,) + (1,) * (d - 1 - k))
    JI = np.linalg.inv(jac)                       # [param (x, y, z order)][physical]
    C = [collocation_derivs_dense(kv, g, 1) for kv, g in zip(kvs, grid)]
    out = apply_tprod_dense([c[0].T for c in C], F[0] * W)
    for a in range(d):                            # parametric direction a (x first) <-> grid axis d-1-a
        Ga = W * sum(JI[..., a, r] * F[1 + r] for r in range(d))
        ops = [C[ax][1 if ax == d - 1 - a else 0].T for ax in range(d)]
        out = out + apply_tprod_dense(ops, Ga)
    return out


def bsp_mixed_deriv_biform_1d(knotvec, du, dv):
    nspans = knotvec.numspans
    nqp = int(math.ceil((2 * knotvec.p - du - dv + 1) / 2.0))
    nodes, qweights = gauss_rule(nqp, knotvec.mesh[:-1], knotvec.mesh[1:])
    derivs = active_deriv(knotvec, nodes, max(du, dv))
    vals1, vals2 = derivs[dv], derivs[du]
    n_act = knotvec.p + 1
    first_act = knotvec.mesh_span_indices() - knotvec.p
    N = knotvec.numdofs
    A = np.zeros((N, N))
    for k in range(nspans):
        f1 = vals1[:, nqp * k:nqp * (k + 1)]
        f2 = vals2[:, nqp * k:nqp * (k + 1)]
        w = qweights[nqp * k:nqp * (k + 1)]
        A[first_act[k]:first_act[k] + n_act, first_act[k]:first_act[k] + n_act] += np.dot(f1, (f2 * w).transpose())
    return scipy.sparse.csr_matrix(A)


def kron_assemble(kind, kvs):
    M = [bsp_mixed_deriv_biform_1d(kv, 0, 0) for kv in kvs]
    k = lambda A, B: scipy.sparse.kron(A, B, format='csr')
    if kind == 'mass':
        out = M[0]
        for X in M[1:]:
            out = k(out, X)
        return out
    K = [bsp_mixed_deriv_biform_1d(kv, 1, 1) for kv in kvs]
    if len(kvs) == 2:
        return k(K[0], M[1]) + k(M[0], K[1])
    M12 = k(M[1], M[2])
    K12 = k(K[1], M[2]) + k(M[1], K[2])
    return k(K[0], M12) + k(M[0], K12)


def read_sparse_matrix(fname):
    """pyiga/utils.py:54-60."""
    I, J, vals = np.loadtxt(fname, skiprows=1, unpack=True)
    I = I.astype(int) - 1
    J = J.astype(int) - 1
    return scipy.sparse.coo_matrix((vals, (I, J))).tocsr()
