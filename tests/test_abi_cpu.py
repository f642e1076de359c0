"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/igx.h declares, the product path fails loudly without a GPU, and the host-side integer
logic (knot vectors, pair enumeration) matches the golden vectors.  No compute calls."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, 'pyiga_amd', 'libigx.so')):
        ge.build()
    import pyiga_amd
    return pyiga_amd._lib


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'igx.h')).read()
    declared = set(re.findall(r'\b(igx_[a-z_0-9]+)\s*\(', hdr))
    declared -= {'igx_ctx', 'igx_patch'}
    bound = {name for name, _, _ in lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    cdll = lib.load()
    for name in declared:
        assert hasattr(cdll, name)
    assert cdll.igx_version() == 101
    nm = subprocess.run(['nm', '-D', '--defined-only', lib.LIB_PATH], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(r'\bT %s\b' % name, nm), name


def test_struct_layout_matches_header(lib, tmp_path):
    """sizeof/offsetof of the ctypes structures equal the C compiler's."""
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "igx.h"\nint main(){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(igx_patch_desc), offsetof(igx_patch_desc, kv),'
                   'offsetof(igx_patch_desc, geo_kind), offsetof(igx_patch_desc, ctrl), offsetof(igx_patch_desc, gauss_x),'
                   'offsetof(igx_patch_desc, row0_lo), offsetof(igx_patch_desc, box_lo), sizeof(igx_patch_info)); printf("%zu\\n", sizeof(igx_timing)); return 0;}')
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split()
    D = lib.PatchDesc
    mine = [ctypes.sizeof(D), D.kv.offset, D.geo_kind.offset, D.ctrl.offset, D.gauss_x.offset, D.row0_lo.offset, D.box_lo.offset,
            ctypes.sizeof(lib.PatchInfo), ctypes.sizeof(lib.Timing)]
    assert [int(x) for x in out] == mine


def test_fails_loudly_without_gpu(lib):
    """No CPU fallback: without a HIP device the product API raises instead of computing."""
    code = ('import sys; sys.path.insert(0, %r)\n'
            'import pyiga_amd\n'
            'from pyiga_amd import bspline, geometry, assemble\n'
            'kv = bspline.make_knots(2, 0.0, 1.0, 4)\n'
            'try:\n'
            '    assemble.stiffness((kv, kv), geometry.unit_square())\n'
            'except pyiga_amd._lib.IgxError as e:\n'
            '    print("RAISED", e)\n' % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES='-1', ROCR_VISIBLE_DEVICES='-1')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env)
    assert 'RAISED' in out.stdout and 'no CPU fallback' in out.stdout, out.stdout + out.stderr


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure; nothing under pyiga_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'pyiga_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'iga_oracle' not in txt and 'from oracle' not in txt and 'import oracle' not in txt, f


def test_host_knot_logic(golden):
    from pyiga_amd import bspline, assemble
    g = golden('bspline')
    for name in sorted({k[:-3] for k in g.files if k.endswith('_kv')}):
        kv = bspline.KnotVector(g[name + '_kv'], int(g[name + '_p']))
        assert np.array_equal(kv.mesh, g[name + '_mesh'])
        assert np.array_equal(kv.mesh_support_idx_all(), g[name + '_meshsupp'])
        assert np.array_equal(assemble._compute_sparsity_ij(kv, kv), g[name + '_sparsity_ij'])
        assert kv.numdofs == g[name + '_C'].shape[0] and kv.numspans == len(g[name + '_mesh']) - 1
    k = golden('knots')
    assert np.array_equal(bspline.make_knots(4, 0.0, 1.0, 128).kv, k['p4_n128'])
    assert np.array_equal(bspline.make_knots(3, 0.0, 1.0, 49).kv, k['p3_n49'])      # the n+1-span quirk, replicated
    assert np.array_equal(bspline.make_knots(2, -1.5, 2.25, 9).kv, k['p2_ab'])
    s = golden('sparsity')
    for name, d in (('d3_p2_n3', 3), ('d2_mixed', 2), ('d3_mult', 3)):
        kvs = tuple(bspline.KnotVector(s['%s_kv%d' % (name, i)], int(s['%s_p%d' % (name, i)])) for i in range(d))
        for lt in (0, 1):
            I, J = assemble._ml_nonzero(kvs, kvs, lower_tri=bool(lt))
            assert np.array_equal(I, s['%s_lt%d_I' % (name, lt)]) and np.array_equal(J, s['%s_lt%d_J' % (name, lt)])


def test_geometry_constructors(golden):
    from pyiga_amd import geometry
    g = golden('geometry')
    cyl = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
    for name, geo in (('quarter_annulus', geometry.quarter_annulus()), ('bspline_quarter_annulus', geometry.bspline_quarter_annulus()),
                      ('twisted_box', geometry.twisted_box()), ('cylinder', cyl),
                      ('unit_square', geometry.unit_square()), ('unit_cube', geometry.unit_cube())):
        assert geo.coeffs.shape == g[name + '_coeffs'].shape, name
        assert np.allclose(geo.coeffs, g[name + '_coeffs'], rtol=0, atol=1e-15), name
        assert isinstance(geo, geometry.NurbsFunc) == bool(g[name + '_nurbs'])
        for k, kv in enumerate(geo.kvs):
            assert kv.p == int(g['%s_gp%d' % (name, k)]) and np.array_equal(kv.kv, g['%s_gkv%d' % (name, k)])
        assert geo.dim == geo.sdim == len(geo.kvs)


def test_shipped_library_has_no_ablation_switches():
    """The switches of the timing experiments are compiled only into -DIGX_ABLATE builds."""
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pyiga_amd', 'libigx.so')
    if not os.path.exists(lib):
        pytest.skip('libigx.so not built')
    blob = open(lib, 'rb').read()
    for name in (b'_DBG', b'IGX_NO_MIRROR', b'IGX_K1PAD', b'IGX_BF_MCHUNKS', b'IGX_FINAL_', b'IGX_FINALQ_', b'IGX_OVERLAP'):
        assert name not in blob, name


def test_bbox_for_rows_matches_reference():
    """Host logic of the on-demand assemblers: the cell box of a set of basis functions equals the one the reference's
    hierarchical discretisation computes (golden_ondemand.npz, made by tests/golden/make_golden.py from the reference)."""
    import numpy as np
    from pyiga_amd import bspline, assemble
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_ondemand.npz'))
    mk = bspline.make_knots
    kvs3 = (mk(2, 0., 1., 6), mk(3, 0., 1., 5), mk(2, 0., 1., 7, mult=2))
    kvs2 = (mk(3, 0., 1., 9), mk(2, 0., 1., 12))
    for name, kvs in (('stiff3d', kvs3), ('convdiff3d', kvs3), ('mass2d', kvs2)):
        assert np.array_equal(np.array(assemble.bbox_for_rows(kvs, g[name + '_rows'])), g[name + '_bbox'])
        I, J = assemble._nonzeros_for_rows(kvs, kvs, g[name + '_rows'])
        assert np.array_equal(J, g[name + '_indices']) and I.size == g[name + '_indptr'][-1]
    assert assemble.bbox_for_rows(kvs2, []) == ((0, 0), (0, 0))


def test_fused_stage_offset_guard(lib):
    """k_bf2 / k_mirror2 address a row block of one outer row and one K1 slice with 32-bit buffer offsets; the host
    predicate that keeps larger patches on the stage kernels (64-bit addresses) -- ADVICE r03: without it the hardware
    range check would drop stores silently."""
    fits = lib.load().igx_fused_stage_fits
    LIM = 0x7fff0000

    def S(N, p):                      # 1D index pairs of an axis with single interior knots
        return N * (2 * p + 1) - p * (p + 1)

    def check(p0, p, N1, N2):
        G1, G2 = (N1 - p) * (p + 1), (N2 - p) * (p + 1)
        want = (2 * p0 + 1) * S(N1, p) * S(N2, p) * 8 <= LIM and G1 * G2 * 8 <= LIM
        assert bool(fits(2 * p0 + 1, S(N1, p), S(N2, p), G1, G2)) == want, (p0, p, N1, N2)
        return want

    assert check(4, 4, 132, 132)                   # C4
    assert check(5, 5, 101, 101)                   # C5
    assert check(2, 2, 66, 66)                     # C3
    assert not check(4, 4, 612, 612)               # a thin slab such as 8 x 608 x 608 spans (nnz below 2^31, row block above)
    assert check(4, 4, 609, 609)
    assert not check(5, 5, 452, 452)
    assert not check(1, 1, 9000, 9000)             # K1 slice beyond 2^31 bytes
    for N in range(560, 640, 7):
        check(4, 4, N, N)
        check(4, 4, N, 2 * N)
    assert fits(1, S(258, 3), S(258, 3), 1024, 1024) == 1      # 2D (C2): one trivial outer row
    assert fits(9, -1, 5, 5, 5) == 0
    assert fits(9, 1 << 40, 1 << 40, 1, 1) == 0    # no overflow of the products


def test_rtc_compile_and_cache(lib, tmp_path, monkeypatch):
    """Run-time compiled coefficient kernels (SURVEY 8 f1 "full": an emitter + an on-disk code-object cache keyed by the
    source hash, the analogue of pyiga/compile.py:58-73,120-132): hiprtc cross-compiles without a GPU.  First call compiles
    into the cache, the second finds the code object there; a different expression gets a different file; a broken
    expression comes back with the compiler's message; the Python front-end translates a restricted expression grammar."""
    cdll = lib.load()
    monkeypatch.setenv('IGX_CACHE_DIR', str(tmp_path / 'cache'))
    buf = ctypes.create_string_buffer(1024)
    hit = ctypes.c_int(-1)
    expr = b'1.0 + x * x + 0.5 * sin(pi * z) - fmax(y, 0.25) / 3.0'
    assert cdll.igx_rtc_compile(expr, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 0
    path = buf.value.decode()
    assert path.startswith(str(tmp_path / 'cache')) and path.endswith('.hsaco')
    assert open(path, 'rb').read(4) == b'\x7fELF' and os.path.getsize(path) > 1000
    assert cdll.igx_rtc_compile(expr, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 1
    assert buf.value.decode() == path
    assert cdll.igx_rtc_compile(b'2.0 * y', b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 0
    assert buf.value.decode() != path
    # a truncated cache file is not trusted: compiled again and replaced
    open(path, 'wb').write(b'\x7fEL')
    assert cdll.igx_rtc_compile(expr, b'gfx950', buf, 1024, ctypes.byref(hit)) == 0 and hit.value == 0
    assert open(path, 'rb').read(4) == b'\x7fELF'
    assert cdll.igx_rtc_compile(b'1.0 + nosuchfunction(x)', b'gfx950', buf, 1024, ctypes.byref(hit)) != 0
    assert b'nosuchfunction' in cdll.igx_last_error()
    # front-end
    from pyiga_amd.assemblers import ExprCoefficient
    e = ExprCoefficient('1 + x**2 + 0.5 * np.sin(pi * z) - maximum(y, 0.25) / 3')
    x, y, z = np.random.default_rng(0).random((3, 50))
    assert np.allclose(e(x, y, z), 1 + x ** 2 + 0.5 * np.sin(np.pi * z) - np.maximum(y, 0.25) / 3, rtol=0, atol=1e-15)
    assert e(0.5, 0.5, 0.5).shape == ()
    assert cdll.igx_rtc_compile(e.c_source().encode(), b'gfx950', buf, 1024, ctypes.byref(hit)) == 0
    for bad in ('__import__("os")', 'x if y else z', 'open("f")', 'x.real', 'lambda: 1', 'q + 1'):
        with pytest.raises((ValueError, SyntaxError)):
            ExprCoefficient(bad)


def test_separable_geometry_detection():
    """geometry.split_axis0 (host only): which control nets are an extruded cross-section along axis 0."""
    from pyiga_amd import geometry as g, bspline
    cyl = g.tensor_product(g.line_segment(0.0, 1.0), g.quarter_annulus())
    zc, g2 = g.split_axis0(cyl)
    assert np.array_equal(zc, [0.0, 1.0]) and isinstance(g2, g.NurbsFunc) and g2.sdim == 2 and g2.dim == 2
    assert np.array_equal(g2.coeffs, g.quarter_annulus().coeffs)
    zc, g2 = g.split_axis0(g.tensor_product(g.line_segment(0.5, 2.0, intervals=3), g.bspline_quarter_annulus()))
    assert np.allclose(zc, [0.5, 1.0, 1.5, 2.0]) and isinstance(g2, bspline.BSplineFunc)
    assert g.split_axis0(g.unit_cube()) is not None
    assert g.split_axis0(g.twisted_box()) is None
    assert g.split_axis0(g.tensor_product(g.quarter_annulus(), g.line_segment(0.0, 1.0))) is None     # separable along the LAST axis only
    assert g.split_axis0(g.quarter_annulus()) is None and g.split_axis0(None) is None
    # a perturbed net is not separable; weights that vary along axis 0 are not recognised
    bad = cyl.copy()
    bad.coeffs[1, 1, 1, 0] += 1e-9
    assert g.split_axis0(bad) is None
    rat = cyl.copy()
    rat.coeffs[1] *= 1.5
    assert g.split_axis0(rat) is None
