#!/bin/bash
# round 4: the coefficient of the convection-diffusion form at C5 size -- sampled on the host and uploaded (a Python
# callable, the reference's way), affine on the device, expression compiled at run time (cold: compile; warm: cache)
cd "$GRAFT_REPO_ROOT"
export IGX_CACHE_DIR=/tmp/igx-rtc-cache; rm -rf $IGX_CACHE_DIR
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coefficient_expression" 2>&1 | tail -3
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from pyiga_amd import bspline, geometry, assemblers
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
kv = bspline.make_knots(5, 0.0, 1.0, 96)
kvs = (kv, kv, kv)
def run(name, coeff):
    t0 = time.perf_counter()
    asm = assemblers.ConvDiffAssembler3D(kvs, geo, coeff)
    asm.patch.ctx.sync()
    t1 = time.perf_counter()
    asm.patch.assemble('convdiff', to_host=False)
    asm.patch.ctx.sync()
    t2 = time.perf_counter()
    d = asm.patch.assemble('convdiff', to_host=False)
    tm = asm.patch.timing()['total_ms']
    print('%-44s set-up %.3f s   first assembly %.3f s   assembly %.2f ms   cache hit: %s' % (name, t1 - t0, t2 - t1, tm, getattr(asm, 'coeff_cache_hit', None)), flush=True)
    asm.patch.close()
run('warm-up (library, context)', assemblers.AffineCoefficient(1.0, 1.0))
run('callable sampled on the host, uploaded', lambda x, y, z: 1.0 + x * x + 0.5 * np.sin(np.pi * z))
run('AffineCoefficient (device)', assemblers.AffineCoefficient(1.0, 1.0))
run('ExprCoefficient, first use (hiprtc compile)', assemblers.ExprCoefficient('1 + x**2 + 0.5 * sin(pi * z)'))
run('ExprCoefficient, second use (disk cache)', assemblers.ExprCoefficient('1 + x**2 + 0.5 * sin(pi * z)'))
PY
