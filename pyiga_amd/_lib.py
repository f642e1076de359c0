"""ctypes binding of libigx (include/igx.h).  Thin: argument marshalling only.

There is no CPU fallback: if the shared library is missing, or no MI355X is
visible when a context is requested, this raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IGX_LIB selects another build of the same library (kernel ablation builds); never a fallback
LIB_PATH = os.environ.get('IGX_LIB') or os.path.join(_HERE, 'libigx.so')

IGX_MASS, IGX_STIFFNESS, IGX_CONVDIFF, IGX_FORM = 0, 1, 2, 3
IGX_GEO_BSPLINE, IGX_GEO_NURBS, IGX_GEO_JACOBIAN = 0, 1, 2
IGX_ALGO_AUTO, IGX_ALGO_ENTRYWISE, IGX_ALGO_SUMFACT = 0, 1, 2
ALGOS = {'auto': IGX_ALGO_AUTO, 'entrywise': IGX_ALGO_ENTRYWISE, 'sumfact': IGX_ALGO_SUMFACT}
KINDS = {'mass': IGX_MASS, 'stiffness': IGX_STIFFNESS, 'convdiff': IGX_CONVDIFF, 'form': IGX_FORM}

_dp = C.POINTER(C.c_double)


class PatchDesc(C.Structure):
    _fields_ = [
        ('dim', C.c_int32), ('p', C.c_int32 * 3), ('kv_len', C.c_int32 * 3), ('kv', _dp * 3),
        ('geo_kind', C.c_int32), ('geo_p', C.c_int32 * 3), ('geo_kv_len', C.c_int32 * 3), ('geo_kv', _dp * 3),
        ('ctrl', _dp), ('jac', _dp),
        ('nqp', C.c_int32), ('gauss_x', _dp), ('gauss_w', _dp),
        ('row0_lo', C.c_int32), ('row0_hi', C.c_int32),
        ('box_lo', C.c_int32 * 3), ('box_hi', C.c_int32 * 3),
    ]


class PatchInfo(C.Structure):
    _fields_ = [
        ('dim', C.c_int32), ('nqp', C.c_int32),
        ('ndofs', C.c_int32 * 3), ('nspans', C.c_int32 * 3), ('ngauss', C.c_int32 * 3),
        ('nrows_total', C.c_int64), ('row_lo', C.c_int64), ('row_hi', C.c_int64),
        ('nnz', C.c_int64), ('nnz_offset', C.c_int64), ('nelem_owned', C.c_int64),
        ('sumfact_ok', C.c_int32), ('reserved', C.c_int32),
    ]


class Timing(C.Structure):
    _fields_ = [
        ('total_ms', C.c_float), ('fields_ms', C.c_float), ('stage0_ms', C.c_float),
        ('stage1_ms', C.c_float), ('final_ms', C.c_float), ('entry_ms', C.c_float),
        ('algo_used', C.c_int32), ('n_launches', C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/igx.h declares: (name, restype, argtypes)
SYMBOLS = [
    ('igx_version', C.c_int, []),
    ('igx_last_error', C.c_char_p, []),
    ('igx_create', C.c_void_p, [C.c_int]),
    ('igx_destroy', None, [C.c_void_p]),
    ('igx_sync', C.c_int, [C.c_void_p]),
    ('igx_stream', C.c_void_p, [C.c_void_p]),
    ('igx_active_deriv', C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, C.c_size_t, C.c_int, _dp]),
    ('igx_find_spans', C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, C.c_size_t, C.POINTER(C.c_int64)]),
    ('igx_grid_jacobian', C.c_int, [C.c_void_p, C.POINTER(PatchDesc), C.c_int, _dp * 3, C.c_int32 * 3, _dp, _dp]),
    ('igx_patch_create', C.c_void_p, [C.c_void_p, C.POINTER(PatchDesc)]),
    ('igx_patch_destroy', None, [C.c_void_p]),
    ('igx_patch_get_info', C.c_int, [C.c_void_p, C.POINTER(PatchInfo)]),
    ('igx_patch_set_coeff', C.c_int, [C.c_void_p, _dp]),
    ('igx_patch_set_form', C.c_int, [C.c_void_p, _dp * 16]),
    ('igx_patch_eval_expr_d', C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    ('igx_patch_set_form_expr', C.c_int, [C.c_void_p, C.c_char_p * 16, C.POINTER(C.c_int)]),
    ('igx_rtc_compile_form', C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    ('igx_patch_form_generated', C.c_int, [C.c_void_p]),
    ('igx_load_vector_expr', C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    ('igx_rtc_compile_load_vector', C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    ('igx_rtc_compile_form_fields', C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    ('igx_patch_set_pform', C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_dp)]),
    ('igx_patch_set_basis_orders', C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ('igx_patch_gauss', C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    ('igx_pattern', C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ('igx_assemble', C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp]),
    ('igx_last_timing', C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    ('igx_d_csr_data', C.c_void_p, [C.c_void_p]),
    ('igx_d_csr_indices', C.c_void_p, [C.c_void_p]),
    ('igx_d_csr_indptr', C.c_void_p, [C.c_void_p]),
    ('igx_entries', C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.c_size_t, _dp]),
    ('igx_fields', C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int64)]),
    ('igx_fused_stage_fits', C.c_int, [C.c_int64] * 5),
    ('igx_assemble_kron3', C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _dp, _dp, _dp]),
    ('igx_patch_set_coeff_expr', C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    ('igx_rtc_compile', C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    ('igx_patch_placement', C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ('igx_load_vector', C.c_int, [C.c_void_p, _dp, _dp]),
    ('igx_load_vector_jet', C.c_int, [C.c_void_p, _dp * 4, _dp]),
    ('igx_load_vector_jet_expr', C.c_int, [C.c_void_p, C.c_char_p * 4, _dp, C.POINTER(C.c_int)]),
    ('igx_entries_d', C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ('igx_load_vector_d', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('igx_fast_assemble', C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    ('igx_patch_set_aca_batch', C.c_int, [C.c_void_p, C.c_longlong]),
    ('igx_fast_assemble_stats', C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    ('igx_patch_set_coeff_affine', C.c_int, [C.c_void_p, C.c_double * 4]),
    ('igx_patch_set_form_d', C.c_int, [C.c_void_p, C.c_void_p * 16]),
    ('igx_patch_last_path', C.c_int, [C.c_void_p]),
    ('igx_patch_gauss_slab', C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ('igx_dev_alloc', C.c_void_p, [C.c_void_p, C.c_size_t]),
    ('igx_dev_free', None, [C.c_void_p, C.c_void_p]),
    ('igx_dev_upload', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ('igx_dev_download', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
]

_lib = None
_lock = threading.Lock()


IGX_OK, IGX_ERR_ARG, IGX_ERR_HIP, IGX_ERR_UNSUPPORTED, IGX_ERR_NOMEM, IGX_ERR_NORTC, IGX_ERR_COMPILE = 0, 1, 2, 3, 4, 5, 6      # include/igx.h
_warned = set()


def sampled_fallback(e, what):
    """May a call that hands an EXPRESSION to the run-time compiler fall back to sampling on the host after error `e`?  Only when
    the device path is not available for this input -- no libhiprtc on the box, the expression does not compile, the kernel does
    not serve the patch -- never after a device failure (IGX_ERR_HIP: failed launch, sticky error) or an allocation failure
    (IGX_ERR_NOMEM): those are re-raised.  The reason is reported once per kind, so that a missing libhiprtc is visible."""
    code = getattr(e, 'code', None)
    if code not in (IGX_ERR_NORTC, IGX_ERR_COMPILE, IGX_ERR_UNSUPPORTED):
        return False
    key = (what, code)
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn('pyiga_amd: %s is sampled on the host (%s)' % (what, str(e).split(': ', 1)[-1][:200]), RuntimeWarning, stacklevel=3)
    return True


class IgxError(RuntimeError):
    """An entry point of libigx returned a status other than IGX_OK; ``code`` carries it (None: raised on the Python side)."""
    code = None


def load():
    """Load libigx.so and declare every prototype.  Raises if the library was not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise IgxError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                               '(or `make -C pyiga_amd/csrc`). pyiga_amd has no CPU fallback.' % LIB_PATH)
            lib = C.CDLL(LIB_PATH)
            for name, res, args in SYMBOLS:
                fn = getattr(lib, name)          # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def last_error():
    return load().igx_last_error().decode('utf-8', 'replace')


def check(rc, what):
    if rc != 0:
        e = IgxError('%s failed (code %d): %s' % (what, rc, last_error()))
        e.code = rc
        raise e


def dptr(a):
    return a.ctypes.data_as(_dp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# -------------------------------------------------------------------------------------------
_ctx = {}


class Context:
    """One per GPU: wraps igx_ctx (device id + HIP stream)."""

    def __init__(self, device=0):
        lib = load()
        self.device = int(device)
        self.handle = lib.igx_create(self.device)
        if not self.handle:
            raise IgxError('igx_create(%d) failed: %s' % (self.device, last_error()))

    def sync(self):
        check(load().igx_sync(self.handle), 'igx_sync')

    def close(self):
        if self.handle:
            load().igx_destroy(self.handle)
            self.handle = None


def default_device():
    """LOCAL_RANK under torch.distributed.run, else IGX_DEVICE, else 0."""
    for key in ('IGX_DEVICE', 'LOCAL_RANK'):
        if key in os.environ:
            return int(os.environ[key])
    return 0


def context(device=None):
    device = default_device() if device is None else int(device)
    with _lock:
        ctx = _ctx.get(device)
    if ctx is None:
        ctx = Context(device)
        with _lock:
            _ctx[device] = ctx
    return ctx


def device_code_sha(path=None):
    """sha256 (first 16 hex digits) of the gfx950 code objects inside the library -- the ``.hip_fatbin`` section of the ELF
    file.  Host-only changes leave it alone: what the committed counter passes (profiles/*traffic.json) were measured on is
    identified by the device code, not by the text of the sources."""
    import hashlib
    import struct
    p = path or LIB_PATH
    with open(p, 'rb') as f:
        eh = f.read(64)
        if eh[:4] != b'\x7fELF' or eh[4] != 2:
            return None
        shoff, = struct.unpack_from('<Q', eh, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from('<HHH', eh, 0x3A)
        f.seek(shoff)
        sh = [struct.unpack_from('<IIQQQQIIQQ', f.read(shentsize)) for _ in range(shnum)]
        f.seek(sh[shstrndx][4])
        names = f.read(sh[shstrndx][5])
        for name_off, _, _, _, off, size, *_ in sh:
            if names[name_off:names.index(b'\0', name_off)] == b'.hip_fatbin':
                f.seek(off)
                return hashlib.sha256(f.read(size)).hexdigest()[:16]
    return None
