"""Assembler objects -- the plugin boundary of the reference.

``MassAssembler{2,3}D`` / ``StiffnessAssembler{2,3}D`` mirror the classes of
``pyiga.assemblers`` (pyiga/assemblers.pyx:26-349,1158-1540; base class
pyiga/genericasm.pxi:312-463,631-786): constructed from ``(kvs, geo)``, exposing ``arity``,
``kvs``, ``entry(i, j)`` and ``multi_entries(indices)``.  All state lives on the MI355X in
an ``igx_patch``; every number is produced by HIP kernels in libigx.

In addition to the reference interface each assembler offers ``assemble_csr()`` which forms
the whole matrix on the device (what ``assemble_entries`` uses) -- the reference has to go
through ``multi_entries`` with one index pair per nonzero, which at 3D p=4 n=128 would be
25 GB of index pairs.
"""
import os
import ctypes as C
import weakref

import numpy as np
import scipy.sparse

from . import _lib
from . import bspline
from .quadrature import make_tensor_quadrature
from . import geometry

_ENTRY_TRAMPOLINES = []          # ctypes trampolines handed out in "entryfunc" capsules (entry_func_ptr)


def _is_spline_geo(geo):
    return hasattr(geo, 'kvs') and hasattr(geo, 'coeffs') and hasattr(geo, 'sdim')


def _is_nurbs(geo):
    from .geometry import NurbsFunc
    return isinstance(geo, NurbsFunc) or type(geo).__name__ == 'NurbsFunc'


class DevicePatch:
    """RAII wrapper of ``igx_patch``: discretisation + geometry resident on one GPU."""

    def __init__(self, kvs, geo, device=None, row0=None, jacobian=None, bbox=None, nqp=None):
        lib = _lib.load()
        self.ctx = _lib.context(device)
        self.kvs = tuple(kvs)
        self.dim = len(self.kvs)
        assert self.dim in (2, 3), 'libigx assembles 2D and 3D patches'
        # Gauss points per span: max p + 1 (pyiga/assemblers.pyx:1338) unless the caller fixes it (a boundary patch uses the
        # rule of the volume space it belongs to)
        self.nqp = int(nqp) if nqp else max(kv.p for kv in self.kvs) + 1
        d = _lib.PatchDesc()
        keep = []
        d.dim = self.dim
        for k, kv in enumerate(self.kvs):
            a = _lib.f64(kv.kv)
            keep.append(a)
            d.kv[k] = _lib.dptr(a)
            d.kv_len[k] = a.size
            d.p[k] = int(kv.p)
        d.nqp = self.nqp
        gx, gw = np.polynomial.legendre.leggauss(self.nqp)      # pyiga/quadrature.py:8
        gx, gw = _lib.f64(gx), _lib.f64(gw)
        keep += [gx, gw]
        d.gauss_x, d.gauss_w = _lib.dptr(gx), _lib.dptr(gw)
        # on-demand assembler (pyiga/codegen/cython.py:541-559): geometry-dependent fields only on the cells
        # bbox[k][0] <= cell < bbox[k][1] of every axis; such a patch serves multi_entries inside the box and nothing else
        self.bbox = None
        if bbox is not None:
            assert row0 is None, 'a bounding box and a row slab exclude each other'
            self.bbox = tuple((int(lo), int(hi)) for lo, hi in bbox)
            assert len(self.bbox) == self.dim, 'bounding box needs one cell range per axis'
            for k, (lo, hi) in enumerate(self.bbox):
                assert 0 <= lo < hi <= self.kvs[k].numspans, 'bounding box outside the mesh (or empty)'
                d.box_lo[k], d.box_hi[k] = lo, hi
        if jacobian is None and _is_spline_geo(geo) and len(geo.kvs) == self.dim:
            d.geo_kind = _lib.IGX_GEO_NURBS if _is_nurbs(geo) else _lib.IGX_GEO_BSPLINE
            for k, gkv in enumerate(geo.kvs):
                a = _lib.f64(gkv.kv)
                keep.append(a)
                d.geo_kv[k] = _lib.dptr(a)
                d.geo_kv_len[k] = a.size
                d.geo_p[k] = int(gkv.p)
            ncomp = self.dim + (1 if d.geo_kind == _lib.IGX_GEO_NURBS else 0)
            ctrl = _lib.f64(geo.coeffs)
            assert ctrl.shape[-1] == ncomp and ctrl.ndim == self.dim + 1, 'control net has wrong shape'
            keep.append(ctrl)
            d.ctrl = _lib.dptr(ctrl)
        else:
            # arbitrary geometry object: take its Jacobians on the Gauss grid as an array
            # (the reference does this for every geometry, assemblers.pyx:1376)
            if jacobian is None:
                meshes = [kv.mesh for kv in self.kvs]
                if self.bbox is not None:                       # (upper cell index exclusive: include its end point)
                    meshes = [m[lo:hi + 1] for m, (lo, hi) in zip(meshes, self.bbox)]
                grid, _ = make_tensor_quadrature(meshes, self.nqp)
                jacobian = geo.grid_jacobian(grid)
            jac = _lib.f64(jacobian)
            G = tuple(kv.numspans * self.nqp for kv in self.kvs)
            if self.bbox is not None:
                G = tuple((hi - lo) * self.nqp for lo, hi in self.bbox)
            assert jac.shape == G + (self.dim, self.dim), 'Jacobian array has wrong shape'
            keep.append(jac)
            d.geo_kind = _lib.IGX_GEO_JACOBIAN
            d.jac = _lib.dptr(jac)
        if row0 is not None:
            d.row0_lo, d.row0_hi = int(row0[0]), int(row0[1])
        self.handle = lib.igx_patch_create(self.ctx.handle, C.byref(d))
        del keep
        if not self.handle:
            raise _lib.IgxError('igx_patch_create failed: ' + _lib.last_error())
        self.info = _lib.PatchInfo()
        _lib.check(lib.igx_patch_get_info(self.handle, C.byref(self.info)), 'igx_patch_get_info')
        self._pattern = None

    def close(self):
        if getattr(self, 'handle', None):
            for name in ('_d_f', '_d_vec', '_d_ij', '_d_val'):
                self._dev_free(name)
            _lib.load().igx_patch_destroy(self.handle)
            self.handle = None

    # -- device-resident buffers (function values, index pairs, results stay in HBM between calls)
    def _dev_free(self, name):
        buf = getattr(self, name, None)
        if buf:
            ctx = getattr(self, 'ctx', None)
            if ctx is not None and getattr(ctx, 'handle', None):      # (the context may be gone at interpreter shutdown)
                _lib.load().igx_dev_free(ctx.handle, buf[0])
            setattr(self, name, None)

    def _dev_buffer(self, name, nbytes):
        buf = getattr(self, name, None)
        if buf and buf[1] >= nbytes:
            return buf[0]
        self._dev_free(name)
        ptr = _lib.load().igx_dev_alloc(self.ctx.handle, nbytes)
        if not ptr:
            raise _lib.IgxError('igx_dev_alloc failed: ' + _lib.last_error())
        setattr(self, name, (ptr, nbytes))
        return ptr

    def resident_grid(self):
        """Shape of the Gauss grid the coefficient arrays of this patch are given on: the whole tensor grid, or the grid of the
        bounding box for an on-demand patch."""
        if self.bbox is not None:
            return tuple((hi - lo) * self.nqp for lo, hi in self.bbox)
        return tuple(self.info.ngauss[k] for k in range(self.dim))

    def gauss_slab(self):
        """(first plane, number of planes) of the Gauss planes of axis 0 that are resident for this row slab."""
        lo, n = C.c_int64(), C.c_int64()
        _lib.check(_lib.load().igx_patch_gauss_slab(self.handle, C.byref(lo), C.byref(n)), 'igx_patch_gauss_slab')
        return int(lo.value), int(n.value)

    def upload_function(self, fvals):
        """Copy the function values (full tensor Gauss grid) of the resident Gauss slab to the device; load_vector_resident()
        then works without any transfer."""
        G = tuple(self.info.ngauss[k] for k in range(self.dim))
        fvals = np.asarray(fvals, dtype=np.float64)
        assert fvals.shape == G, 'function values have the wrong grid shape'
        g0_lo, g0_n = self.gauss_slab()
        part = np.ascontiguousarray(fvals[g0_lo:g0_lo + g0_n])
        ptr = self._dev_buffer('_d_f', part.nbytes)
        _lib.check(_lib.load().igx_dev_upload(self.ctx.handle, ptr, part.ctypes.data, part.nbytes), 'igx_dev_upload')

    def eval_function_expr(self, c_expr, parametric=False):
        """The function values of load_vector_resident() from a C expression in x, y, z, evaluated on the device at the resident
        Gauss points (physical coordinates, or the parametric ones) by a kernel compiled at run time (igx_patch_eval_expr_d):
        nothing is sampled on the host or uploaded.  Returns True if the code object came from the cache."""
        g0_lo, g0_n = self.gauss_slab()
        npts = g0_n * int(np.prod([self.info.ngauss[k] for k in range(1, self.dim)]))
        ptr = self._dev_buffer('_d_f', 8 * npts)
        hit = C.c_int(0)
        _lib.check(_lib.load().igx_patch_eval_expr_d(self.handle, c_expr.encode(), 1 if parametric else 0, ptr, C.byref(hit)), 'igx_patch_eval_expr_d')
        return bool(hit.value)

    def load_vector_expr(self, c_expr, parametric=False):
        """Load vector of a scalar function given as a C expression in x, y, z (physical coordinates, or the parametric ones).  3D
        patches the fused contraction kernel serves: a generated variant of it evaluates the function at the points of its grid
        line (igx_load_vector_expr) -- the function values never exist as an array; anything else: the function values by a
        generated kernel, then the contractions (igx_patch_eval_expr_d + igx_load_vector_d)."""
        lo, hi = int(self.info.row_lo), int(self.info.row_hi)
        nd = self.ndofs
        out = np.empty(((hi - lo) // int(np.prod(nd[1:])),) + nd[1:])
        hit = C.c_int(0)
        rc = _lib.load().igx_load_vector_expr(self.handle, c_expr.encode(), 1 if parametric else 0, _lib.dptr(out), C.byref(hit))
        if rc == _lib.IGX_ERR_UNSUPPORTED:
            self.eval_function_expr(c_expr, parametric=parametric)
            return self.load_vector_resident(to_host=True)
        _lib.check(rc, 'igx_load_vector_expr')
        return out

    def load_vector_resident(self, to_host=False):
        """Load vector from the function values uploaded with upload_function(); the result stays on the device unless asked for."""
        assert getattr(self, '_d_f', None), 'upload_function() first'
        lo, hi = int(self.info.row_lo), int(self.info.row_hi)
        nd = self.ndofs
        n0 = (hi - lo) // int(np.prod(nd[1:]))
        n_out = n0 * int(np.prod(nd[1:]))
        d_out = self._dev_buffer('_d_vec', 8 * n_out)
        _lib.check(_lib.load().igx_load_vector_d(self.handle, self._d_f[0], d_out), 'igx_load_vector_d')
        if not to_host:
            return None
        out = np.empty((n0,) + nd[1:])
        _lib.check(_lib.load().igx_dev_download(self.ctx.handle, out.ctypes.data, d_out, out.nbytes), 'igx_dev_download')
        return out

    def upload_pairs(self, idx):
        """Index pairs (M, 2) of a batched multi_entries request, kept on the device for entries_resident()."""
        idx = np.ascontiguousarray(idx, dtype=np.uintp)
        assert idx.ndim == 2 and idx.shape[1] == 2
        ptr = self._dev_buffer('_d_ij', idx.nbytes)
        _lib.check(_lib.load().igx_dev_upload(self.ctx.handle, ptr, idx.ctypes.data, idx.nbytes), 'igx_dev_upload')
        self._npairs = idx.shape[0]

    def entries_resident(self, kind, to_host=False):
        assert getattr(self, '_d_ij', None), 'upload_pairs() first'
        M = self._npairs
        d_val = self._dev_buffer('_d_val', 8 * M)
        _lib.check(_lib.load().igx_entries_d(self.handle, _lib.KINDS[kind], self._d_ij[0], M, d_val), 'igx_entries_d')
        if not to_host:
            return None
        out = np.empty(M)
        _lib.check(_lib.load().igx_dev_download(self.ctx.handle, out.ctypes.data, d_val, out.nbytes), 'igx_dev_download')
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- sizes
    @property
    def ndofs(self):
        return tuple(self.info.ndofs[k] for k in range(self.dim))

    @property
    def shape(self):
        n = int(self.info.nrows_total)
        return (n, n)

    @property
    def nnz(self):
        return int(self.info.nnz)

    @property
    def row_range(self):
        return int(self.info.row_lo), int(self.info.row_hi)

    # -- operations
    def pattern(self):
        """(indptr, indices) of the owned rows, canonical CSR, int32 (local indptr)."""
        if self._pattern is None:
            nrows = self.row_range[1] - self.row_range[0]
            indptr = np.empty(nrows + 1, dtype=np.int32)
            indices = np.empty(self.nnz, dtype=np.int32)
            _lib.check(_lib.load().igx_pattern(self.handle, indptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                               indices.ctypes.data_as(C.POINTER(C.c_int32))), 'igx_pattern')
            self._pattern = (indptr, indices)
        return self._pattern

    def assemble(self, kind, algo='auto', to_host=True):
        """Form all CSR values on the device; returns them (host array) if `to_host`."""
        out = np.empty(self.nnz) if to_host else None
        _lib.check(_lib.load().igx_assemble(self.handle, _lib.KINDS[kind], _lib.ALGOS[algo],
                                            _lib.dptr(out) if to_host else None), 'igx_assemble')
        return out

    def assemble_kron(self, kind, patch2, m0, k0=None, to_host=True):
        """Mass / stiffness values of this 3D patch from the 2D patch of its cross-section and the weighted 1D matrices of
        axis 0 (igx_assemble_kron3: separable geometry, see ``geometry.split_axis0`` and ``assemble.separable_terms``)."""
        out = np.empty(self.nnz) if to_host else None
        m0 = _lib.f64(m0)
        k0 = None if k0 is None else _lib.f64(k0)
        _lib.check(_lib.load().igx_assemble_kron3(self.handle, patch2.handle, _lib.KINDS[kind], _lib.dptr(m0),
                                                  None if k0 is None else _lib.dptr(k0), _lib.dptr(out) if to_host else None),
                   'igx_assemble_kron3')
        return out

    def fast_assemble(self, kind, tol=1e-10, maxiter=100, skipcount=3, tolcount=3, verbose=0, batch=None):
        """Low-rank (ACA) assembly (igx_fast_assemble): scipy CSR, plus the number of crosses, of evaluated entries and of
        batched device requests in ``self.aca_stats``.  `batch`: slices (3D) / matrices (2D) of the reordered tensor with at
        most that many entries are fetched exactly in one request (default 65536; 0 = line by line like the reference)."""
        assert self.row_range == (0, self.shape[0]), 'the low-rank assembler works on whole patches'
        lib = _lib.load()
        if batch is not None:
            _lib.check(lib.igx_patch_set_aca_batch(self.handle, int(batch)), 'igx_patch_set_aca_batch')
        data = np.empty(self.nnz)
        rank, nent, nreq = C.c_int(0), C.c_longlong(0), C.c_longlong(0)
        _lib.check(lib.igx_fast_assemble(self.handle, _lib.KINDS[kind], float(tol), int(maxiter), int(skipcount), int(tolcount),
                                         int(verbose), _lib.dptr(data), C.byref(rank), C.byref(nent)), 'igx_fast_assemble')
        _lib.check(lib.igx_fast_assemble_stats(self.handle, C.byref(nreq), None, None), 'igx_fast_assemble_stats')
        self.aca_stats = {'rank': rank.value, 'entries': nent.value, 'requests': nreq.value, 'nnz': int(self.nnz)}
        indptr, indices = self.pattern()
        return scipy.sparse.csr_matrix((data, indices, indptr), shape=self.shape)

    def last_path(self):
        """Kernels of the last sum-factorised assembly: set of 'geoA', 'fused', 'both' (the fused stage wrote both triangles), 'bf3' (the fused stage ran as k_bf3), 'mirror', 'single', 'twin' (the chain ran on the patch with mid and last axis exchanged) (include/igx.h IGX_PATH_*)."""
        bits = _lib.load().igx_patch_last_path(self.handle)
        return {name for bit, name in ((1, 'geoA'), (2, 'fused'), (4, 'mirror'), (8, 'single'), (16, 'kron'), (32, 'both'), (64, 'bf3'), (128, 'twin')) if bits & bit}

    def placement(self):
        """Outcome of the opt-in buffer placement search (IGX_PLACEMENT_TRIES, include/igx.h igx_patch_placement):
        {'tried': n, 'best_ms': .., 'worst_ms': ..}; tried == 0: the search did not run."""
        n, b, w = C.c_int(0), C.c_float(0), C.c_float(0)
        _lib.check(_lib.load().igx_patch_placement(self.handle, C.byref(n), C.byref(b), C.byref(w)), 'igx_patch_placement')
        return {'tried': n.value, 'best_ms': round(b.value, 4), 'worst_ms': round(w.value, 4)}

    def timing(self):
        t = _lib.Timing()
        _lib.check(_lib.load().igx_last_timing(self.handle, C.byref(t)), 'igx_last_timing')
        return t.as_dict()

    def entries(self, kind, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uintp)
        assert idx.ndim == 2 and idx.shape[1] == 2
        out = np.zeros(idx.shape[0])
        _lib.check(_lib.load().igx_entries(self.handle, _lib.KINDS[kind], idx.ctypes.data_as(C.POINTER(C.c_size_t)),
                                           idx.shape[0], _lib.dptr(out)), 'igx_entries')
        return out

    def fields(self, kind):
        """Quadrature fields, shape (F, G0_local, G1[, G2]) (W, or the upper triangle of B)."""
        shp = (C.c_int64 * 4)()
        _lib.check(_lib.load().igx_fields(self.handle, _lib.KINDS[kind], None, shp), 'igx_fields')
        shape = tuple(int(x) for x in shp)[:1 + self.dim]
        out = np.empty(shape)
        _lib.check(_lib.load().igx_fields(self.handle, _lib.KINDS[kind], _lib.dptr(out), shp), 'igx_fields')
        return out

    def set_coeff(self, values):
        """Scalar coefficient on the tensor Gauss grid (of the bounding box, for an on-demand patch) for the convection-diffusion form."""
        G = self.resident_grid()
        c = _lib.f64(np.broadcast_to(values, G))
        _lib.check(_lib.load().igx_patch_set_coeff(self.handle, _lib.dptr(c)), 'igx_patch_set_coeff')

    def set_coeff_affine(self, c):
        """Coefficient c[0] + c[1] x + c[2] y + c[3] z of the physical point, evaluated on the device."""
        arr = (C.c_double * 4)(*[float(v) for v in c])
        _lib.check(_lib.load().igx_patch_set_coeff_affine(self.handle, arr), 'igx_patch_set_coeff_affine')

    def set_coeff_expr(self, c_expr):
        """Coefficient given as a C expression in x, y, z (ExprCoefficient.c_source): compiled at run time for the device
        (hiprtc, code object cached on disk by source hash) and evaluated on the resident Gauss points.  Returns True when
        the code object came from the cache."""
        hit = C.c_int(0)
        _lib.check(_lib.load().igx_patch_set_coeff_expr(self.handle, c_expr.encode(), C.byref(hit)), 'igx_patch_set_coeff_expr')
        return bool(hit.value)

    def load_vector(self, fvals):
        """Inner products of the owned basis functions with a function given by its values on the full
        tensor Gauss grid (scalar: shape G; vector-valued: G + trailing component axes)."""
        G = tuple(self.info.ngauss[k] for k in range(self.dim))
        fvals = np.asarray(fvals, dtype=np.float64)
        assert fvals.shape[:self.dim] == G, 'function values have the wrong grid shape'
        lo, hi = int(self.info.row_lo), int(self.info.row_hi)
        nd = self.ndofs
        n0 = (hi - lo) // int(np.prod(nd[1:]))
        comp_shape = fvals.shape[self.dim:]
        ncomp = int(np.prod(comp_shape)) if comp_shape else 1
        flat = fvals.reshape(G + (ncomp,))
        out = np.empty((n0,) + nd[1:] + (ncomp,))
        lib = _lib.load()
        for c in range(ncomp):
            fc = _lib.f64(flat[..., c])
            oc = np.empty((n0,) + nd[1:])
            _lib.check(lib.igx_load_vector(self.handle, _lib.dptr(fc), _lib.dptr(oc)), 'igx_load_vector')
            out[..., c] = oc
        return out.reshape((n0,) + nd[1:] + comp_shape)

    def set_form(self, table):
        """Physical coefficient table of IGX_FORM: 4x4 nested list of arrays on the Gauss grid (see resident_grid) or None."""
        G = self.resident_grid()
        keep, ptrs = [], (_lib._dp * 16)()
        for r in range(4):
            for s in range(4):
                e = table[r][s]
                if e is not None:
                    arr = _lib.f64(np.broadcast_to(e, G))
                    keep.append(arr)
                    ptrs[4 * r + s] = _lib.dptr(arr)
        if not keep:
            raise ValueError('the form has no non-zero coefficient')
        _lib.check(_lib.load().igx_patch_set_form(self.handle, ptrs), 'igx_patch_set_form')

    def form_generated(self):
        """True if the form set last is served by a generated field kernel: no coefficient arrays exist on the device."""
        return bool(_lib.load().igx_patch_form_generated(self.handle))

    def set_form_expr(self, table):
        """Coefficient table of IGX_FORM as C expressions in x, y, z (4x4 nested list of strings or None): the FIELD kernel of the
        form is generated with the expressions inside (geometry, coefficients, jet transformation in one pass), compiled for the
        device at run time and cached on disk (igx_patch_set_form_expr).  Returns True if the code object came from the cache."""
        exprs = (C.c_char_p * 16)()
        for r in range(4):
            for s in range(4):
                if table[r][s] is not None:
                    exprs[4 * r + s] = table[r][s].encode()
        hit = C.c_int(0)
        _lib.check(_lib.load().igx_patch_set_form_expr(self.handle, exprs, C.byref(hit)), 'igx_patch_set_form_expr')
        return bool(hit.value)

    def set_basis_orders(self, slot0=None, slot1=None):
        """Derivative orders held by the two slots of every axis' basis table (default (0, 1): value and first derivative); with
        other orders the patch assembles parametric jet forms only (igx_patch_set_basis_orders)."""
        d = self.dim
        s0 = (C.c_int * 3)(*(tuple(slot0) if slot0 is not None else (0,) * d), *([0] * (3 - d)))
        s1 = (C.c_int * 3)(*(tuple(slot1) if slot1 is not None else (1,) * d), *([1] * (3 - d)))
        _lib.check(_lib.load().igx_patch_set_basis_orders(self.handle, s0, s1), 'igx_patch_set_basis_orders')

    def set_pform(self, terms):
        """Parametric jet form of IGX_FORM: list of (mask_v, mask_u, coefficient array on the full Gauss grid) -- at most 16
        terms, masks over the grid axes (igx_patch_set_pform)."""
        G = self.resident_grid()
        n = len(terms)
        masks = (C.c_int * (2 * n))()
        ptrs = (_lib._dp * n)()
        keep = []
        for k, (mv, mu, c) in enumerate(terms):
            masks[2 * k], masks[2 * k + 1] = int(mv), int(mu)
            arr = _lib.f64(np.broadcast_to(c, G))
            keep.append(arr)
            ptrs[k] = _lib.dptr(arr)
        _lib.check(_lib.load().igx_patch_set_pform(self.handle, n, masks, ptrs), 'igx_patch_set_pform')

    def load_vector_jet(self, jet):
        """Load vector of  sum_r F_r D_r v  (jet[0]: coefficient of v, jet[1..d]: of its physical derivatives;
        arrays on the full Gauss grid or None)."""
        G = tuple(self.info.ngauss[k] for k in range(self.dim))
        keep, ptrs = [], (_lib._dp * 4)()
        for r, e in enumerate(jet):
            if e is not None:
                arr = _lib.f64(np.broadcast_to(e, G))
                keep.append(arr)
                ptrs[r] = _lib.dptr(arr)
        lo, hi = int(self.info.row_lo), int(self.info.row_hi)
        nd = self.ndofs
        out = np.empty(((hi - lo) // int(np.prod(nd[1:])),) + nd[1:])
        _lib.check(_lib.load().igx_load_vector_jet(self.handle, ptrs, _lib.dptr(out)), 'igx_load_vector_jet')
        return out

    def load_vector_jet_expr(self, exprs):
        """The same with the coefficients as C expressions in x, y, z (strings or None), compiled for the device at run time
        (igx_load_vector_jet_expr)."""
        arr = (C.c_char_p * 4)()
        for r, e in enumerate(exprs):
            if e is not None:
                arr[r] = e.encode()
        lo, hi = int(self.info.row_lo), int(self.info.row_hi)
        nd = self.ndofs
        out = np.empty(((hi - lo) // int(np.prod(nd[1:])),) + nd[1:])
        hit = C.c_int(0)
        _lib.check(_lib.load().igx_load_vector_jet_expr(self.handle, arr, _lib.dptr(out), C.byref(hit)), 'igx_load_vector_jet_expr')
        return out

    def gauss(self, axis):
        """Gauss nodes and weights of an axis (of its part inside the bounding box, for an on-demand patch)."""
        n = self.info.ngauss[axis]
        nodes, weights = np.empty(n), np.empty(n)
        _lib.check(_lib.load().igx_patch_gauss(self.handle, axis, _lib.dptr(nodes), _lib.dptr(weights)), 'igx_patch_gauss')
        if self.bbox is not None:
            lo, hi = self.bbox[axis]
            return nodes[lo * self.nqp:hi * self.nqp], weights[lo * self.nqp:hi * self.nqp]
        return nodes, weights

    def csr(self, kind, algo='auto'):
        """scipy CSR of the owned rows (all rows for a full patch): f64 data, int32 indices."""
        data = self.assemble(kind, algo=algo, to_host=True)
        indptr, indices = self.pattern()
        nrows = self.row_range[1] - self.row_range[0]
        return scipy.sparse.csr_matrix((data, indices, indptr), shape=(nrows, self.shape[1]))


class _DeviceAssembler:
    """Common part of the four assembler classes (genericasm.pxi BaseAssembler{2,3}D)."""
    _kind = None
    _dim = None
    arity = 2

    @classmethod
    def inputs(cls):
        return {'geo': (cls._dim,)}

    @classmethod
    def parameters(cls):
        return {}

    def __init__(self, kvs0, geo, device=None, row0=None, bbox=None):
        assert len(kvs0) == self._dim, 'Assembler requires %d knot vectors' % self._dim
        assert geo.sdim == self._dim, 'Geometry has wrong source dimension'
        assert geo.dim == self._dim, 'Geometry has wrong dimension'
        self._geo = geo
        self.nqp = max(kv.p for kv in kvs0) + 1
        kvs0 = tuple(kvs0)
        self.kvs = (kvs0, kvs0)
        # bbox: the reference's on-demand assemblers (compile_vform(..., on_demand=True), pyiga/_hdiscr.py:37-56) -- cell
        # ranges per axis outside of which nothing is precomputed; multi_entries then serves pairs inside the box
        self.patch = DevicePatch(kvs0, geo, device=device, row0=row0, bbox=bbox)

    # --- the reference's per-entry interface (genericasm.pxi:677-758)
    def entry(self, i, j):
        """One matrix entry for the ravelled dof indices (i, j)."""
        return float(self.patch.entries(self._kind, np.array([[i, j]], dtype=np.uintp))[0])

    def entry_func_ptr(self):
        """PyCapsule named "entryfunc" holding a C function ``double(*)(size_t i, size_t j, void *data)`` that returns
        entry (i, j) (pyiga/genericasm.pxi:780-786; consumed by pyiga/fast_assemble_cy.pyx:102-103, which passes the
        assembler object as `data`).  The function is a ctypes trampoline into :meth:`entry` -- one device call per entry:
        for compatibility only; batched consumers should use :meth:`multi_entries` or the ACA assembler of this package
        (``DevicePatch.fast_assemble``)."""
        if getattr(self, '_entry_cb', None) is None:
            me = weakref.proxy(self)                           # no self -> callback -> self cycle
            proto = C.CFUNCTYPE(C.c_double, C.c_size_t, C.c_size_t, C.c_void_p)
            self._entry_cb = proto(lambda i, j, _data: float(me.entry(int(i), int(j))))
            # a capsule may outlive the assembler: the trampoline stays registered for the life of the process (a call
            # after the assembler is gone raises ReferenceError inside the callback instead of jumping to freed memory)
            _ENTRY_TRAMPOLINES.append(self._entry_cb)
        new = C.pythonapi.PyCapsule_New
        new.restype, new.argtypes = C.py_object, [C.c_void_p, C.c_char_p, C.c_void_p]
        return new(C.cast(self._entry_cb, C.c_void_p), b'entryfunc', None)

    def multi_entries(self, indices):
        """All entries for an ``N x 2`` array (or iterable) of (row, col) pairs; pairs whose
        supports do not intersect give 0.0."""
        if isinstance(indices, np.ndarray):
            idx = np.asarray(indices, order='C', dtype=np.uintp)
        else:
            idx = np.array(list(indices), dtype=np.uintp)
        return self.patch.entries(self._kind, idx.reshape(-1, 2))

    def entry1(self, i):
        return 0.0                       # arity 2: same answer as the reference

    def multi_entries1(self, indices):
        return None

    def assemble_vector(self):
        return None

    # --- whole-matrix path
    def assemble_csr(self, algo='auto'):
        return self.patch.csr(self._kind, algo=algo)


class MassAssembler2D(_DeviceAssembler):
    _kind, _dim = 'mass', 2


class StiffnessAssembler2D(_DeviceAssembler):
    _kind, _dim = 'stiffness', 2


class MassAssembler3D(_DeviceAssembler):
    _kind, _dim = 'mass', 3


class StiffnessAssembler3D(_DeviceAssembler):
    _kind, _dim = 'stiffness', 3


class AffineCoefficient:
    """c(x, y, z) = c0 + c1 x + c2 y + c3 z in physical coordinates.  Callable like any coefficient function (so it also
    works with the reference); the device assemblers recognise it and evaluate it on the GPU instead of sampling it on
    the host and shipping one double per Gauss point."""

    def __init__(self, c0, c1=0.0, c2=0.0, c3=0.0):
        self.c = (float(c0), float(c1), float(c2), float(c3))

    def __call__(self, x, y, z):
        return self.c[0] + self.c[1] * x + self.c[2] * y + self.c[3] * z


class ExprCoefficient:
    """A coefficient function given as an expression string in the physical coordinates, e.g.
    ``ExprCoefficient('1 + x**2 + 0.5 * sin(pi * z)')``.  Callable like any coefficient function -- numpy evaluates the
    expression, so it also works with the reference -- and the device assemblers recognise it: the expression is translated
    to C, compiled for the GPU at run time and evaluated there on the Gauss points (``igx_patch_set_coeff_expr``: the
    counterpart of the reference's run-time compiled assemblers, pyiga/compile.py:58-73), instead of being sampled on the
    host and shipped as one double per Gauss point.

    Grammar: numbers, ``x y z pi``, ``+ - * / **``, unary minus, and the functions sin cos tan exp log sqrt tanh sinh cosh
    abs minimum maximum power (numpy names; also min / max / pow / fabs)."""

    _FUNCS = {'sin': 'sin', 'cos': 'cos', 'tan': 'tan', 'exp': 'exp', 'log': 'log', 'sqrt': 'sqrt', 'tanh': 'tanh', 'sinh': 'sinh',
              'cosh': 'cosh', 'abs': 'fabs', 'fabs': 'fabs', 'minimum': 'fmin', 'maximum': 'fmax', 'min': 'fmin', 'max': 'fmax',
              'power': 'pow', 'pow': 'pow', 'arctan': 'atan', 'atan': 'atan', 'arctan2': 'atan2', 'atan2': 'atan2'}
    _ARITY = {'fmin': 2, 'fmax': 2, 'pow': 2, 'atan2': 2}

    def __init__(self, expr):
        import ast
        self.expr = str(expr)
        self._tree = ast.parse(self.expr, mode='eval')
        self._c = self._emit(self._tree.body)          # validates the grammar

    def _emit(self, n):
        import ast
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)) and not isinstance(n.value, bool):
            return repr(float(n.value))
        if isinstance(n, ast.Name) and n.id in ('x', 'y', 'z', 'pi'):
            return n.id
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.USub, ast.UAdd)):
            return '(%s%s)' % ('-' if isinstance(n.op, ast.USub) else '+', self._emit(n.operand))
        if isinstance(n, ast.BinOp):
            a, b = self._emit(n.left), self._emit(n.right)
            ops = {ast.Add: '+', ast.Sub: '-', ast.Mult: '*', ast.Div: '/'}
            if type(n.op) in ops:
                return '(%s %s %s)' % (a, ops[type(n.op)], b)
            if isinstance(n.op, ast.Pow):
                if isinstance(n.right, ast.Constant) and n.right.value in (2, 3, 4):        # small integer powers as products (as numpy does)
                    return '(' + ' * '.join([a] * int(n.right.value)) + ')'
                return 'pow(%s, %s)' % (a, b)
        if isinstance(n, ast.Call) and not n.keywords:
            f = n.func
            name = f.id if isinstance(f, ast.Name) else f.attr if (isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in ('np', 'numpy', 'math')) else None
            if name in self._FUNCS:
                cf = self._FUNCS[name]
                if len(n.args) == self._ARITY.get(cf, 1):
                    return '%s(%s)' % (cf, ', '.join(self._emit(a) for a in n.args))
        raise ValueError('ExprCoefficient: unsupported construct in %r' % self.expr)

    def c_source(self):
        return self._c

    def __call__(self, x, y, z):
        ns = {k: getattr(np, k) for k in ('sin', 'cos', 'tan', 'exp', 'log', 'sqrt', 'tanh', 'sinh', 'cosh', 'abs', 'fabs', 'minimum',
                                          'maximum', 'power', 'arctan', 'arctan2')}
        ns.update(min=np.minimum, max=np.maximum, pow=np.power, atan=np.arctan, atan2=np.arctan2, np=np, numpy=np, math=np, pi=np.pi,
                  x=np.asarray(x, dtype=float), y=np.asarray(y, dtype=float), z=np.asarray(z, dtype=float))
        return eval(compile(self._tree, '<ExprCoefficient>', 'eval'), {'__builtins__': {}}, ns) + 0.0 * ns['x']


class ConvDiffAssembler3D(_DeviceAssembler):
    """Assembler for the variational form

        (inner(diff_coeff*grad(u),grad(v)) + inner((x[1],-x[0],1.0),grad(u))*v)*dx

    -- what the reference compiles at run time from that string (pyiga/assemble.py:837-897,
    pyiga/vform.py, pyiga/codegen/cython.py).  Non-symmetric.  `diff_coeff` is a function of the
    physical coordinates (x, y, z), evaluated on the Gauss grid through the geometry exactly like
    ``pyiga.utils.grid_eval_transformed`` (host numpy, as in the reference); everything else runs on
    the device.
    """
    _symmetric_form = False
    _kind, _dim = 'convdiff', 3

    @classmethod
    def inputs(cls):
        return {'geo': (3,), 'diff_coeff': ()}

    def __init__(self, kvs0, geo, diff_coeff, device=None, row0=None, bbox=None):
        super().__init__(kvs0, geo, device=device, row0=row0, bbox=bbox)
        if isinstance(diff_coeff, AffineCoefficient) and isinstance(geo, (bspline.BSplineFunc, geometry.NurbsFunc)):
            self.patch.set_coeff_affine(diff_coeff.c)             # evaluated on the device: nothing sampled on the host
            return
        # (run-time compilation needs libhiprtc on the box and can fail on an expression: a failure falls through to the sampled
        # coefficient below -- same matrix, set-up on the host)
        if isinstance(diff_coeff, ExprCoefficient) and isinstance(geo, (bspline.BSplineFunc, geometry.NurbsFunc)) and bbox is None:
            try:
                self.coeff_cache_hit = self.patch.set_coeff_expr(diff_coeff.c_source())     # compiled for the device at run time
                return
            except _lib.IgxError as e:
                if not _lib.sampled_fallback(e, 'the coefficient of the convection-diffusion form'):
                    raise
        if callable(diff_coeff) and isinstance(geo, (bspline.BSplineFunc, geometry.NurbsFunc)) and bbox is None \
                and os.environ.get('IGX_FORM_RTC', '1') != '0':
            # a plain Python callable: traced into a C expression (pyiga_amd.symbolic) and compiled like an ExprCoefficient;
            # what cannot be traced (comparisons, np.where, ...) is sampled on the host below
            from . import symbolic
            try:
                X = symbolic.coordinates(3)
                src = symbolic.c_source(np.broadcast_to(np.asarray(diff_coeff(X[..., 0], X[..., 1], X[..., 2]), dtype=object), (1, 1, 1)))
            except Exception:
                src = None
            if src is not None:
                try:
                    self.coeff_cache_hit = self.patch.set_coeff_expr(src)
                    self.coeff_traced = True
                    return
                except _lib.IgxError as e:
                    if not _lib.sampled_fallback(e, 'the coefficient of the convection-diffusion form'):
                        raise
        # (any other geometry object: the coefficient is sampled through geo.grid_eval like a plain callable)
        grid = [self.patch.gauss(k)[0] for k in range(3)]
        X = geo.grid_eval(grid)                                   # shape(grid) x 3, components (x, y, z)
        vals = diff_coeff(X[..., 0], X[..., 1], X[..., 2]) if callable(diff_coeff) else diff_coeff
        self.patch.set_coeff(vals)


class _GeneralFormAssembler(_DeviceAssembler):
    """Scalar bilinear form in the first-order jets of u and v,

        a(u, v) = integral of  sum_{r,s=0..d} P_rs(x) D_r v D_s u  dx,   D_0 = id, D_1..d = d/dx, d/dy[, d/dz],

    given either as a form string in the reference's syntax (``'(inner(dot(K, grad(u)), grad(v)) + c*u*v) * dx'``,
    evaluated by ``pyiga_amd.forms``) or directly as a (d+1)x(d+1) table of coefficient functions/arrays.  This
    is the class of forms the reference's run-time compiler handles with ``u``, ``v``, ``grad``, ``inner``, ``dot``
    (pyiga/assemble.py:837-897, pyiga/vform.py:1804-1885); non-symmetric, every pattern entry is computed.
    Coefficients are sampled on the Gauss grid on the host (they are Python callables, as in the reference),
    the Jacobian transformation of the coefficients and all sums run on the device (sum-factorised stages
    or entry-wise kernel).
    """
    _symmetric_form = False
    _kind = 'form'

    def __init__(self, kvs0, geo, form, inputs=None, device=None, row0=None, bbox=None):
        from . import forms
        super().__init__(kvs0, geo, device=device, row0=row0, bbox=bbox)
        d = self._dim
        # (1) the coefficients as generated device code: the string and its callable inputs are traced into C expressions in
        # the physical coordinates, one kernel per form is compiled at run time and cached (the reference: one compiled module
        # per form, pyiga/compile.py:58-73) -- nothing is sampled on the host.  IGX_FORM_RTC=0 switches it off.
        self.compiled, self.coeff_cache_hit = False, None
        if isinstance(form, str) and bbox is None and os.environ.get('IGX_FORM_RTC', '1') != '0' \
                and isinstance(geo, (bspline.BSplineFunc, geometry.NurbsFunc)):
            try:
                traced = forms.symbolic_table(form, d, dict(inputs or {}))
            except NotImplementedError:
                raise                                             # the form itself is outside this front-end
            except Exception:
                traced = None                                     # an input that cannot be traced: sampled below
            if traced is not None:
                full = [[None] * 4 for _ in range(4)]
                for r in range(d + 1):
                    for s in range(d + 1):
                        full[r][s] = traced[r][s]
                if not any(e is not None for row in full for e in row):
                    raise ValueError('the form has no non-zero coefficient')
                try:
                    self.coeff_cache_hit = self.patch.set_form_expr(full)
                    self.table_mask = [[e is not None for e in row] for row in traced]
                    self.compiled = True
                    return
                except _lib.IgxError as e:                        # no hiprtc on this box / compile error: sampled below
                    if not _lib.sampled_fallback(e, 'the coefficient table of a form'):
                        raise
        # (2) coefficients sampled on the Gauss grid on the host
        grid = [self.patch.gauss(k)[0] for k in range(d)]
        G = tuple(len(g) for g in grid)
        X = np.asarray(geo.grid_eval(grid))                       # shape(grid) x d, components (x, y[, z])
        if isinstance(form, str):
            table = forms.coefficient_table(form, G, X, dict(inputs or {}))
        else:
            table = [[None] * (d + 1) for _ in range(d + 1)]
            for r in range(d + 1):
                for s in range(d + 1):
                    e = form[r][s]
                    if e is not None:
                        table[r][s] = np.broadcast_to(e(*(X[..., k] for k in range(d))) if callable(e) else e, G)
        full = [[None] * 4 for _ in range(4)]
        for r in range(d + 1):
            for s in range(d + 1):
                full[r][s] = table[r][s]
        self.table_mask = [[e is not None for e in row] for row in table]
        self.patch.set_form(full)


class _ParametricFormAssembler(_DeviceAssembler):
    """Scalar bilinear forms with second derivatives and / or parametric derivatives of u and v -- ``hess``, ``Dx(., k, times=2)``,
    ``div(grad(.))``, ``grad(., parametric=True)`` (pyiga/vform.py:1518-1600; transformation of physical second derivatives:
    pyiga/vform.py:592-625).  ``pyiga_amd.pforms`` turns the string into a parametric jet form (coefficient arrays on the Gauss
    grid, the geometry factors worked out on the host); the device assembles it in passes -- the basis tables of a pass hold the
    two derivative orders per axis its terms need (``pforms.plan_passes``) -- and the passes are added up here."""
    _symmetric_form = False
    _kind = 'form'

    def __init__(self, kvs0, geo, form, inputs=None, device=None, row0=None):
        from . import pforms
        super().__init__(kvs0, geo, device=device, row0=row0)
        d = self._dim
        grid = [self.patch.gauss(k)[0] for k in range(d)]
        G = tuple(len(g) for g in grid)
        X = np.asarray(geo.grid_eval(grid))
        Jac = np.asarray(geo.grid_jacobian(grid))
        if hasattr(geo, 'parametric_derivatives'):
            H2 = geo.parametric_derivatives(grid)[2]
        else:
            from .bspline import _hessian_index_pairs
            Hl = np.asarray(geo.grid_hessian(grid))               # shape(grid) x d x num_hess, linearised upper triangle
            H2 = np.empty(G + (d, d, d))
            for n, (i, j) in enumerate(_hessian_index_pairs(d)):
                H2[..., i, j] = H2[..., j, i] = Hl[..., n]
        self.terms = pforms.evaluate(form, G, X, Jac, H2, dict(inputs or {}))
        self.passes = pforms.plan_passes(self.terms, d)

    def _run(self, call):
        """Sum of call() over the passes, the basis tables set for each; the default tables are restored afterwards."""
        total = None
        try:
            for slot0, slot1, terms in self.passes:
                self.patch.set_basis_orders(slot0, slot1)
                self.patch.set_pform(terms)
                r = call()
                total = r if total is None else total + r
        finally:
            self.patch.set_basis_orders()
        return total

    def multi_entries(self, indices):
        idx = (np.asarray(indices, order='C', dtype=np.uintp) if isinstance(indices, np.ndarray)
               else np.array(list(indices), dtype=np.uintp)).reshape(-1, 2)
        return self._run(lambda: self.patch.entries('form', idx))

    def entry(self, i, j):
        return float(self.multi_entries(np.array([[i, j]], dtype=np.uintp))[0])

    def assemble_csr(self, algo='auto'):
        data = self._run(lambda: self.patch.assemble('form', algo=algo))
        indptr, indices = self.patch.pattern()
        nrows = self.patch.row_range[1] - self.patch.row_range[0]
        return scipy.sparse.csr_matrix((data, indices, indptr), shape=(nrows, self.patch.shape[1]))


class ParametricFormAssembler2D(_ParametricFormAssembler):
    _dim = 2


class ParametricFormAssembler3D(_ParametricFormAssembler):
    _dim = 3


class GeneralFormAssembler2D(_GeneralFormAssembler):
    _dim = 2


class GeneralFormAssembler3D(_GeneralFormAssembler):
    _dim = 3


class _FunctionalAssembler:
    """Arity-1 assemblers: the load vector  (f, v)  over the patch (pyiga/assemblers.pyx:883-1156,
    2204-2500; base class genericasm.pxi:312-463,631-786).  `f` is sampled on the Gauss grid on the host
    (it is a Python callable or a spline function, as in the reference); weights, |det J| and the sum
    over the Gauss grid run on the device."""
    _dim = None
    _physical = False
    arity = 1

    @classmethod
    def inputs(cls):
        return {'geo': (cls._dim,), 'f': ()}

    @classmethod
    def parameters(cls):
        return {}

    def __init__(self, kvs0, geo, f, device=None, row0=None):
        from . import utils
        assert len(kvs0) == self._dim, 'Assembler requires %d knot vectors' % self._dim
        assert geo.sdim == self._dim, 'Geometry has wrong source dimension'
        assert geo.dim == self._dim, 'Geometry has wrong dimension'
        self._geo = geo
        kvs0 = tuple(kvs0)
        self.kvs = (kvs0,)
        self.nqp = max(kv.p for kv in kvs0) + 1
        self.patch = DevicePatch(kvs0, geo, device=device, row0=row0)
        self.gaussgrid = tuple(self.patch.gauss(k)[0] for k in range(self._dim))
        self._vector = None
        self._f = f
        # a plain callable is traced into a C expression and evaluated on the device (pyiga_amd.symbolic); spline functions
        # and whatever cannot be traced are sampled on the Gauss grid on the host
        from . import symbolic
        self._fexpr = symbolic.trace_function(f, self._dim) if (not self._physical or isinstance(geo, (bspline.BSplineFunc, geometry.NurbsFunc))) else None
        if self._fexpr is not None:
            self._fvals = None
        elif self._physical:
            self._fvals = utils.grid_eval_transformed(f, self.gaussgrid, geo)
        else:
            self._fvals = utils.grid_eval(f, self.gaussgrid)

    def assemble_vector(self):
        if self._vector is None:
            if self._fexpr is not None:
                try:
                    self._vector = self.patch.load_vector_expr(self._fexpr, parametric=not self._physical)
                except _lib.IgxError as e:                       # (no run-time compiler on this box: sampled on the host)
                    if not _lib.sampled_fallback(e, 'the function of a load vector'):
                        raise
                    self._fexpr = None
                    self._fvals = utils.grid_eval_transformed(self._f, self.gaussgrid, self._geo) if self._physical else utils.grid_eval(self._f, self.gaussgrid)
            if self._vector is None:
                self._vector = self.patch.load_vector(self._fvals)
        return self._vector.copy()

    # per-entry interface of the reference (genericasm.pxi:353-436,677-758), arity 1
    def entry1(self, i):
        return float(self.assemble_vector().ravel()[i])

    def multi_entries1(self, indices):
        idx = np.asarray(list(indices) if not isinstance(indices, np.ndarray) else indices, dtype=np.intp)
        return self.assemble_vector().ravel()[idx.ravel()]

    def entry(self, i, j):
        return self.entry1(i)           # arity 1: the column index is ignored, as in the reference

    def multi_entries(self, indices):
        return None


class _FormFunctionalAssembler(_FunctionalAssembler):
    """Linear functional given as a form string, e.g. ``'f * v * dx'`` or ``'(f * v + inner(b, grad(v))) * dx'``
    (pyiga/assemble.py:837-897 with arity 1): the coefficients of v and grad(v) (inputs, ``x``, numbers) are
    evaluated on the Gauss grid by ``pyiga_amd.forms``; weights, the Jacobian transformation and the sums
    run on the device."""
    _physical = True

    def __init__(self, kvs0, geo, form, inputs=None, device=None, row0=None):
        from . import forms
        super().__init__(kvs0, geo, lambda *xyz: 0.0, device=device, row0=row0)
        self._form, self._inputs = form, inputs
        # the coefficients of v and grad(v) as generated device code when the string and its inputs can be traced (pyiga_amd.symbolic)
        self._jet_exprs = None
        if os.environ.get('IGX_FORM_RTC', '1') != '0' and isinstance(geo, (bspline.BSplineFunc, geometry.NurbsFunc)):
            try:
                from . import symbolic
                traced = forms.functional_jet(form, (1,) * self._dim, symbolic.coordinates(self._dim), dict(inputs or {}), traced=True)
                self._jet_exprs = [None if e is None else symbolic.c_source(e) for e in traced]
            except NotImplementedError:
                raise
            except Exception:
                self._jet_exprs = None
        if self._jet_exprs is not None:
            self._jet = [None if e is None else True for e in self._jet_exprs]
            return
        G = tuple(len(g) for g in self.gaussgrid)
        X = np.asarray(geo.grid_eval(list(self.gaussgrid)))
        self._jet = forms.functional_jet(form, G, X, dict(inputs or {}))

    def assemble_vector(self):
        if self._vector is None:
            if self._jet_exprs is not None and any(e is not None for e in self._jet_exprs):
                try:
                    self._vector = self.patch.load_vector_jet_expr(self._jet_exprs + [None] * (4 - len(self._jet_exprs)))
                    return self._vector.copy()
                except _lib.IgxError as e:                       # (no run-time compiler on this box: the jet is sampled on the host)
                    if not _lib.sampled_fallback(e, 'the jet of a functional'):
                        raise
                    from . import forms
                    G = tuple(len(g) for g in self.gaussgrid)
                    self._jet = forms.functional_jet(self._form, G, np.asarray(self._geo.grid_eval(list(self.gaussgrid))), dict(self._inputs or {}))
                    self._jet_exprs = None
            if all(e is None for e in self._jet):
                lo, hi = self.patch.row_range                # the zero functional ('0 * v * dx')
                nd = self.patch.ndofs
                self._vector = np.zeros(((hi - lo) // int(np.prod(nd[1:])),) + tuple(nd[1:]))
            elif all(e is None for e in self._jet[1:]):
                self._vector = self.patch.load_vector(self._jet[0])
            else:
                self._vector = self.patch.load_vector_jet(self._jet)
        return self._vector.copy()


class GeneralFunctionalAssembler2D(_FormFunctionalAssembler):
    _dim = 2


class GeneralFunctionalAssembler3D(_FormFunctionalAssembler):
    _dim = 3


class L2FunctionalAssembler2D(_FunctionalAssembler):
    _dim = 2


class L2FunctionalAssembler3D(_FunctionalAssembler):
    _dim = 3


class L2FunctionalAssemblerPhys2D(_FunctionalAssembler):
    _dim, _physical = 2, True


class L2FunctionalAssemblerPhys3D(_FunctionalAssembler):
    _dim, _physical = 3, True
