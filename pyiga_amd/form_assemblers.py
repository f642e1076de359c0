"""Assemblers for form strings with vector-valued basis functions and for boundary integrals.

What the reference compiles per form (pyiga/vform.py, pyiga/codegen/cython.py) and drives through ``assemble_entries_vec``
/ ``multi_blocks`` (pyiga/assemble.py:760-811, pyiga/genericasm.pxi:158-235) is expressed here as a set of scalar jet
forms -- one per pair of components -- that the device assembles with ``IGX_FORM`` (sum-factorised or entry-wise kernels):

* volume integrals: the blocks share one device patch (geometry, tables); only the coefficient table changes per block;
* boundary integrals (``... * ds`` with ``boundary=``): the trace space has one function along the normal axis (its value 1
  and its normal derivative at the face enter the jets, pyiga/codegen/cython.py:566-570), so the form becomes a
  (d-1)-dimensional jet form on the face.  The geometry algebra of the face (unit normal, surface measure, physical
  gradients from tangential and normal parametric derivatives) is a few small numpy operations on the face's Gauss grid;
  the sums run on the device on a (d-1)-dimensional patch with the identity map (3D problems) or, for the 1D faces of 2D
  problems, on the host like every 1D matrix of this package (pyiga_amd/assemble.py, row a3).
"""
import functools

import numpy as np
import scipy.sparse

from . import bspline, geometry, tforms
from .assemblers import DevicePatch
from .quadrature import make_tensor_quadrature


def parse_bdspec(bdspec, dim):
    """(axis, side) of a boundary name or pair (pyiga/bspline.py:13-33)."""
    names = {'left': (dim - 1, 0), 'right': (dim - 1, 1), 'bottom': (dim - 2, 0), 'top': (dim - 2, 1),
             'front': (dim - 3, 0), 'back': (dim - 3, 1)}
    bd = names.get(bdspec, bdspec) if isinstance(bdspec, str) else bdspec
    if isinstance(bd, str) or not (len(bd) == 2 and bd[1] in (0, 1)):
        raise ValueError('invalid bdspec ' + str(bd))
    if bd[0] < 0 or bd[0] >= dim:
        raise ValueError('invalid bdspec %s for space of dimension %d' % (bdspec, dim))
    return int(bd[0]), int(bd[1])


def _identity_geo(kvs):
    segs = [geometry.line_segment(kv.support()[0], kv.support()[1], support=kv.support()) for kv in kvs]
    return functools.reduce(geometry.tensor_product, segs)


def _full_table(tab, d):
    """(d+1)x(d+1) table -> the 4x4 table DevicePatch.set_form takes."""
    full = [[None] * 4 for _ in range(4)]
    for r in range(d + 1):
        for s in range(d + 1):
            full[r][s] = tab[r][s]
    return full


def _assemble_1d_jets(kv, nqp, C=None, F=None):
    """1D jet form on the host: A[i][j] = sum_g w_g sum_ab C[a][b](g) D_a v_i(g) D_b u_j(g)   (or the vector with F[a])."""
    nodes, weights = make_tensor_quadrature([kv.mesh], nqp)
    first, vals = bspline.collocation_derivs_info(kv, nodes[0], derivs=1)           # (2, G, p+1)
    Gn, N = nodes[0].size, kv.numdofs
    B = np.zeros((2, Gn, N))
    cols = first[:, None] + np.arange(kv.p + 1)[None, :]
    for a in range(2):
        B[a][np.arange(Gn)[:, None], cols] = vals[a]
    w = weights[0]
    if F is not None:
        out = np.zeros(N)
        for a in range(2):
            if F[a] is not None:
                out += B[a].T @ (w * F[a])
        return out
    A = np.zeros((N, N))
    for a in range(2):
        for b in range(2):
            if C[a][b] is not None:
                A += B[a].T @ ((w * C[a][b])[:, None] * B[b])
    return scipy.sparse.csr_matrix(A)


class FormAssembler:
    """Assembler object for a form string with `bfuns` (vector-valued functions) and/or `boundary`; the reference's
    plugin interface for such forms: ``arity``, ``kvs``, ``num_components()``, ``multi_blocks()``, ``assemble_vector()``
    plus ``assemble(format, layout)``, which ``assemble.assemble_entries`` calls."""

    def __init__(self, kvs, geo, form, bfuns=None, inputs=None, boundary=None, device=None):
        kvs = tuple(kvs)
        self.kvs = (kvs, kvs)
        self._geo = geo
        self._form = form
        self._device = device
        d = len(kvs)
        assert geo.sdim == d, 'Geometry has wrong source dimension'
        self.nqp = max(kv.p for kv in kvs) + 1
        inputs = {k: v for k, v in dict(inputs or {}).items() if k != 'geo'}
        self.boundary = None if boundary is None else parse_bdspec(boundary, d)
        self._surface = boundary is None and geo.dim == d + 1
        assert self._surface or geo.dim == d, 'Geometry has wrong dimension'
        if self._surface:
            self._setup_surface(kvs, geo, form, bfuns, inputs)
        elif self.boundary is None:
            assert d in (2, 3), 'vector-valued forms are assembled for 2D and 3D patches'
            self._tkvs = kvs
            self.patch = DevicePatch(kvs, geo, device=device)
            grid = [self.patch.gauss(k)[0] for k in range(d)]
            G = tuple(len(g) for g in grid)
            X = np.asarray(geo.grid_eval(grid))
            self.arity, measure, table, self._ncs = tforms.evaluate(form, G, X, inputs, bfuns)
            if measure != 'dx':
                raise ValueError('a surface integral (ds) needs the `boundary` argument')
            self._table = table
        else:
            self._setup_boundary(kvs, geo, form, bfuns, inputs)
        # (all functions scalar: the reference forces a scalar assembler even when `bfuns` spells them out, pyiga/vform.py:1854-1856)
        self._vector_valued = any(nc > 1 for nc in self._ncs)

    # ---- surface integrals: a d-dimensional patch mapped into (d+1)-dimensional space (pyiga/vform.py:46-54,205-211,1839-1845)
    def _setup_surface(self, kvs, geo, form, bfuns, inputs):
        d = len(kvs)
        self._tkvs = kvs
        nodes, _ = make_tensor_quadrature([kv.mesh for kv in kvs], self.nqp)
        G = tuple(len(g) for g in nodes)
        X = np.asarray(geo.grid_eval(list(nodes))).reshape(G + (d + 1,))
        Jac = np.asarray(geo.grid_jacobian(list(nodes))).reshape(G + (d + 1, d))
        if d == 1:
            un = np.stack([-Jac[..., 1, 0], Jac[..., 0, 0]], axis=-1)
        else:
            un = np.cross(Jac[..., :, 0], Jac[..., :, 1])
        ds = np.sqrt(np.sum(un * un, axis=-1))
        self.arity, measure, table, self._ncs = tforms.evaluate(form, G, X, inputs, bfuns, normal=un / ds[..., None])
        if measure != 'ds':
            raise ValueError('an integral over a surface patch must be written with ds')

        def value_only(jet):
            if any(e is not None for e in jet[1:]):
                raise NotImplementedError('derivatives of basis functions in a surface integral')
            return None if jet[0] is None else np.ascontiguousarray(jet[0] * ds)
        if self.arity == 2:
            def block(tab):
                if any(tab[r][s] is not None for r in range(d + 2) for s in range(d + 2) if r or s):
                    raise NotImplementedError('derivatives of basis functions in a surface integral')
                C = [[None] * (d + 1) for _ in range(d + 1)]
                C[0][0] = None if tab[0][0] is None else np.ascontiguousarray(tab[0][0] * ds)
                return C
            self._table = [[block(tab) for tab in row] for row in table]
        else:
            self._table = [[value_only(jet)] + [None] * d for jet in table]
        self.patch = DevicePatch(kvs, _identity_geo(kvs), device=self._device, nqp=self.nqp) if d == 2 else None

    # ---- boundary integrals
    def _setup_boundary(self, kvs, geo, form, bfuns, inputs):
        d = len(kvs)
        ax, side = self.boundary
        tkvs = tuple(kv for k, kv in enumerate(kvs) if k != ax)
        self._tkvs = tkvs
        nkv = kvs[ax]
        xi_n = nkv.support()[side]
        # value / derivative of the one basis function that lives on the face (first or last of the open knot vector)
        nd = bspline.active_deriv(nkv, np.array([xi_n]), 1)                  # (2, p+1, 1)
        k_act = 0 if side == 0 else nkv.p
        assert abs(nd[0, k_act, 0] - 1.0) < 1e-13, 'boundary integrals need an open knot vector'
        c_n = float(nd[1, k_act, 0])
        tnodes, _ = make_tensor_quadrature([kv.mesh for kv in tkvs], self.nqp)
        grid = list(tnodes)
        grid.insert(ax, np.array([xi_n]))
        G = tuple(len(g) for g in tnodes)
        X = np.asarray(geo.grid_eval(grid)).reshape(G + (d,))
        Jac = np.asarray(geo.grid_jacobian(grid)).reshape(G + (d, d))       # [..., i, j] = d x_i / d xi_j, xi in (x, y, z) order
        Jinv = np.linalg.inv(Jac)                                            # [..., j, i] = d xi_j / d x_i
        det = np.linalg.det(Jac)
        jn = d - 1 - ax                                                      # the normal parametric direction in (x, y, z) order
        # unscaled outer normal: the cross product of the face's tangents = det(J) grad(xi_n), outward for a positively
        # oriented patch (the reference's convention, pyiga/assemble.py:898-911)
        un = (1.0 if side else -1.0) * det[..., None] * Jinv[..., jn, :]
        ds = np.sqrt(np.sum(un * un, axis=-1))
        normal = un / ds[..., None]
        self.arity, measure, table, self._ncs = tforms.evaluate(form, G, X, inputs, bfuns, normal=normal)
        if measure != 'ds':
            raise ValueError('a boundary integral must be written with ds')
        # physical jets of a trace function from the jets of its (d-1)-dimensional factor: D_r phi = sum_a M[r][a] d_a phi_t
        taxes = [k for k in range(d) if k != ax]
        jx = [d - 1 - k for k in reversed(taxes)]                           # parametric directions of the face patch, (x, y) order
        M = np.zeros(G + (d + 1, d))
        M[..., 0, 0] = 1.0
        for i in range(d):
            M[..., i + 1, 0] = c_n * Jinv[..., jn, i]
            for a, j in enumerate(jx):
                M[..., i + 1, a + 1] = Jinv[..., j, i]
        self._M, self._ds = M, ds

        def to_param(tab):                 # bilinear block: C[a][b] = ds * sum_rs M[r][a] P[r][s] M[s][b]
            C = [[None] * d for _ in range(d)]
            for a in range(d):
                for b in range(d):
                    acc = None
                    for r in range(d + 1):
                        for s in range(d + 1):
                            if tab[r][s] is not None:
                                t = M[..., r, a] * tab[r][s] * M[..., s, b]
                                acc = t if acc is None else acc + t
                    if acc is not None and np.any(acc != 0.0):
                        C[a][b] = np.ascontiguousarray(acc * ds)
            return C

        def to_param1(jet):
            F = [None] * d
            for a in range(d):
                acc = None
                for r in range(d + 1):
                    if jet[r] is not None:
                        t = M[..., r, a] * jet[r]
                        acc = t if acc is None else acc + t
                if acc is not None and np.any(acc != 0.0):
                    F[a] = np.ascontiguousarray(acc * ds)
            return F
        if self.arity == 2:
            self._table = [[to_param(blk) for blk in row] for row in table]
        else:
            self._table = [to_param1(jet) for jet in table]
        self.patch = None
        if d == 3:
            self.patch = DevicePatch(tkvs, _identity_geo(tkvs), device=self._device, nqp=self.nqp)

    # ---- the reference's interface
    def num_components(self):
        """(components of the trial function, of the test function), like pyiga/genericasm.pxi:158-159."""
        return (self._ncs[0], self._ncs[-1]) if self.arity == 2 else (self._ncs[0], 1)

    @property
    def _jd(self):
        return len(self._tkvs)            # jets of the assembled patch: value + this many derivatives

    def _block_matrix(self, tab):
        n = int(np.prod([kv.numdofs for kv in self._tkvs]))
        if all(e is None for row in tab for e in row):
            return scipy.sparse.csr_matrix((n, n))
        if self.patch is None:
            return _assemble_1d_jets(self._tkvs[0], self.nqp, C=tab)
        self.patch.set_form(_full_table(tab, self._jd))
        return self.patch.csr('form')

    def blocks(self):
        """[[A_pq]]: scalar matrices of test component p against trial component q."""
        assert self.arity == 2
        return [[self._block_matrix(tab) for tab in row] for row in self._table]

    def multi_blocks(self, indices):
        """Blocks of the entries (i, j): row-major (test component p, trial component q), i.e. block[p][q] =
        A_packed[i * ncv + p, j * ncu + q] for square blocks (pyiga/genericasm.pxi:177-212)."""
        assert self.arity == 2
        idx = np.asarray(indices if isinstance(indices, np.ndarray) else list(indices), dtype=np.uintp).reshape(-1, 2)
        ncv, ncu = self._ncs[1], self._ncs[0]
        out = np.zeros((idx.shape[0], ncv, ncu))
        for p in range(ncv):
            for q in range(ncu):
                tab = self._table[p][q]
                if all(e is None for row in tab for e in row):
                    continue
                if self.patch is None:
                    A = _assemble_1d_jets(self._tkvs[0], self.nqp, C=tab)
                    out[:, p, q] = np.asarray(A[idx[:, 0], idx[:, 1]]).ravel()
                else:
                    self.patch.set_form(_full_table(tab, self._jd))
                    out[:, p, q] = self.patch.entries('form', idx)
        # the reference declares the result as N x numcomp[0] x numcomp[1] (trial, test) over the same row-major
        # (test, trial) memory: identical for square blocks, a reinterpretation for rectangular ones
        return out.reshape(idx.shape[0], ncu, ncv)

    def multi_entries(self, indices):
        assert self.arity == 2 and self._ncs == (1, 1), 'multi_entries is for scalar forms: use multi_blocks'
        return self.multi_blocks(indices)[:, 0, 0]

    def entry(self, i, j):
        return float(self.multi_entries(np.array([[i, j]], dtype=np.uintp))[0])

    def assemble_vector(self):
        """ndofs + (components,) array of the linear functional (components axis also for scalar functions declared through
        `bfuns`, like the reference's vector-valued assemblers; plain scalar forms: ndofs)."""
        assert self.arity == 1
        d = len(self.kvs[0])
        shape = [kv.numdofs for kv in self.kvs[0]]
        if self.boundary is not None:
            shape[self.boundary[0]] = 1                  # one function along the normal axis (pyiga/codegen/cython.py:566-570)
        comps = []
        for jet in self._table:
            if all(e is None for e in jet):
                comps.append(np.zeros(shape))
            elif self.patch is None:
                comps.append(_assemble_1d_jets(self._tkvs[0], self.nqp, F=jet).reshape(shape))
            else:
                full = list(jet) + [None] * (4 - len(jet))
                comps.append(np.asarray(self.patch.load_vector_jet(full)).reshape(shape))
        if not self._vector_valued:
            return comps[0]
        return np.stack(comps, axis=-1)

    def assemble(self, format='csr', layout='blocked'):
        """Matrix in the reference's layouts (pyiga/assemble.py:703-811): 'blocked' = ncv x ncu block matrix of scalar matrices,
        'packed' = every entry a small ncv x ncu block (format 'bsr' keeps the blocks)."""
        if self.arity == 1:
            res = self.assemble_vector()
            return np.moveaxis(res, -1, 0) if self._vector_valued and layout == 'blocked' else res
        B = self.blocks()
        ncv, ncu = self._ncs[1], self._ncs[0]
        if not self._vector_valued:
            return B[0][0].asformat(format)
        if layout == 'blocked':
            return scipy.sparse.bmat(B, format='csr').asformat(format)
        if layout != 'packed':
            raise ValueError('layout must be blocked or packed')
        A = None
        for p in range(ncv):
            for q in range(ncu):
                E = scipy.sparse.csr_matrix(([1.0], ([p], [q])), shape=(ncv, ncu))
                t = scipy.sparse.kron(B[p][q], E, format='csr')
                A = t if A is None else A + t
        if format == 'bsr':
            return scipy.sparse.bsr_matrix(A, blocksize=(ncv, ncu))
        return A.asformat(format)
