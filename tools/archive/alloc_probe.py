#!/usr/bin/env python3
"""Do the kernel times of the C4 chain depend on where the big buffers land?  Several patches in ONE process (each allocates
its own K1 / CSR buffers), optionally with a dummy allocation in between that shifts the addresses."""
import sys, os, ctypes
os.environ.setdefault('IGX_STAGE_EVENTS', '1')      # per-kernel times (off by default below 2^24 Gauss points)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyiga_amd import bspline, geometry, assemblers, _lib

kv = bspline.make_knots(4, 0.0, 1.0, 128)
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
ctx = _lib.context()
pads = [0, 0, 1 << 20, 0, 3 << 20, 0, 64 << 10, 0]
for rep, pad in enumerate(pads):
    dummy = _lib.load().igx_dev_alloc(ctx.handle, pad) if pad else None
    patch = assemblers.DevicePatch((kv, kv, kv), geo)
    for _ in range(2):
        patch.assemble('stiffness', to_host=False)
    ts = []
    for _ in range(4):
        patch.assemble('stiffness', to_host=False)
        t = patch.timing()
        ts.append((t['stage0_ms'], t['stage1_ms'], t['final_ms']))
    m = np.median(np.array(ts), axis=0)
    lib = _lib.load()
    lib.igx_d_csr_data.restype = ctypes.c_void_p
    addr = lib.igx_d_csr_data(patch.handle)
    print('rep %d pad %8d  geoA %.3f  bf2 %.3f  mirror %.3f   csr data at 0x%x' % (rep, pad, m[0], m[1], m[2], addr or 0), flush=True)
    patch.close()
    if dummy:
        _lib.load().igx_dev_free(ctx.handle, dummy)
