#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import os, time, numpy as np
os.environ.setdefault('IGX_STAGE_EVENTS', '1')      # per-kernel times (off by default below 2^24 Gauss points)
import pyiga_amd as iga
for p, n in ((5, 64), (3, 96), (2, 128)):
    kv = iga.bspline.make_knots(p, 0., 1., n)
    geo = iga.geometry.tensor_product(iga.geometry.line_segment(0.0, 1.0), iga.geometry.quarter_annulus())
    patch = iga.assemblers.DevicePatch((kv,) * 3, geo)
    for env in ({}, {'IGX_GEOA': '0'}, {'IGX_GEOA': '0', 'IGX_PATH': 'unfused'}):
        for k in ('IGX_GEOA', 'IGX_PATH'):
            os.environ.pop(k, None)
        os.environ.update(env)
        ts = []
        for _ in range(4):
            patch.assemble('stiffness', algo='sumfact', to_host=False)
            ts.append(patch.timing())
        t = ts[-1]
        print('p=%d n=%d %-40s total %.3f ms  fields %.3f  A %.3f  B/bf %.3f  final/mirror %.3f  path %s' % (p, n, env, t['total_ms'], t['fields_ms'], t['stage0_ms'], t['stage1_ms'], t['final_ms'], sorted(patch.last_path())))
    patch.close()
PY
