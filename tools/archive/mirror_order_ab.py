#!/usr/bin/env python3
"""A/B of launch-time switches of the ablation build inside ONE process and ONE patch (the mirror pass follows where the
driver put the CSR buffer: different processes differ by +-0.4 ms, one patch is stable): alternate the values of an
environment variable between assemblies.  usage: IGX_LIB=pyiga_amd/libigx_ablate.so python3 tools/mirror_order_ab.py VAR v0 v1 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyiga_amd import bspline, geometry, assemblers

var, vals = sys.argv[1], sys.argv[2:]
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
kv = bspline.make_knots(4, 0.0, 1.0, 128)
patch = assemblers.DevicePatch((kv, kv, kv), geo)
for _ in range(3):
    patch.assemble('stiffness', to_host=False)
res = {v: [] for v in vals}
for rep in range(4):
    for v in vals:
        os.environ[var] = v
        for _ in range(3):
            patch.assemble('stiffness', to_host=False)
            t = patch.timing()
            res[v].append((t['stage0_ms'], t['stage1_ms'], t['final_ms'], t['total_ms']))
for v in vals:
    m = np.median(np.array(res[v]), axis=0)
    print('%s=%s: geoA %.3f bf2 %.3f mirror %.3f chain %.3f ms (median of %d)' % (var, v, m[0], m[1], m[2], m[3], len(res[v])))
