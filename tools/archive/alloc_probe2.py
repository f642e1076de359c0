#!/usr/bin/env python3
"""Several patches ALIVE at once in one process (each with its own 12.75 GB CSR buffer and 17 GB K1): do their mirror / fused-stage
times differ, i.e. does the physical placement of a buffer decide them?"""
import sys, os
os.environ.setdefault('IGX_STAGE_EVENTS', '1')      # per-kernel times (off by default below 2^24 Gauss points)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyiga_amd import bspline, geometry, assemblers, _lib
import ctypes

kv = bspline.make_knots(4, 0.0, 1.0, 128)
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
patches = []
for rep in range(int(os.environ.get("NPATCH", "6"))):
    patch = assemblers.DevicePatch((kv, kv, kv), geo)
    patches.append(patch)
    for _ in range(2):
        patch.assemble('stiffness', to_host=False)
for rnd in range(2):
    for rep, patch in enumerate(patches):
        ts = []
        for _ in range(4):
            patch.assemble('stiffness', to_host=False)
            t = patch.timing()
            ts.append((t['stage0_ms'], t['stage1_ms'], t['final_ms']))
        m = np.median(np.array(ts), axis=0)
        lib = _lib.load()
        lib.igx_d_csr_data.restype = ctypes.c_void_p
        addr = lib.igx_d_csr_data(patch.handle) or 0
        print('round %d patch %d  geoA %.3f  bf2 %.3f  mirror %.3f   csr at 0x%x (GiB %.3f, mod 2^30: 0x%x)' % (rnd, rep, m[0], m[1], m[2], addr, addr / 2**30, addr % 2**30), flush=True)
