#!/usr/bin/env python3
"""k_bf2 time of 3D stiffness patches of several degrees for the library given by IGX_LIB (variant builds with other
tile / contractor shapes: tools/buildvar.sh).  usage: IGX_LIB=... python3 tools/shape_try.py"""
import sys, os
os.environ.setdefault('IGX_STAGE_EVENTS', '1')      # per-kernel times (off by default below 2^24 Gauss points)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyiga_amd import bspline, geometry, assemblers

geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
out = []
for p, n in [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or ((1, 96), (2, 64), (3, 64), (4, 48)):
    kv = bspline.make_knots(p, 0.0, 1.0, n)
    for kind in (os.environ.get("KIND", "stiffness"),):
        patch = assemblers.DevicePatch((kv, kv, kv), geo)
        for _ in range(3):
            patch.assemble(kind, to_host=False)
        ts = []
        for _ in range(8):
            patch.assemble(kind, to_host=False)
            t = patch.timing()
            ts.append((t['stage0_ms'], t['stage1_ms'], t['final_ms']))
        m = np.median(np.array(ts), axis=0)
        out.append('p=%d n=%d %-9s geoA %.3f bf2 %.3f mirror %.3f' % (p, n, kind, m[0], m[1], m[2]))
        patch.close()
print(os.path.basename(os.environ.get('IGX_LIB', 'libigx.so')))
print('\n'.join(out))
