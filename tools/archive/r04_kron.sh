#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "separable" 2>&1 | tail -4
for c in c4 c3; do timeout 300 python bench.py --op structured --config $c --steps 10 2>&1 | tail -1 > gpurun_out/r04_${c}_structured_bench.json; python -c "
import json; d = json.load(open('gpurun_out/r04_${c}_structured_bench.json')); print('$c', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['config']['note'])"; done
python - <<'PY'
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from pyiga_amd import bspline, geometry, assemble
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
kv = bspline.make_knots(4, 0.0, 1.0, 128)
for sep in ('1', '0'):
    os.environ['IGX_SEPARABLE'] = sep
    t0 = time.perf_counter(); A = assemble.stiffness((kv, kv, kv), geo); dt = time.perf_counter() - t0
    print('assemble.stiffness at C4, IGX_SEPARABLE=%s: %.2f s end to end (nnz %d), row sums %.2e' % (sep, dt, A.nnz, abs(A @ np.ones(A.shape[0])).max() / abs(A.data).max()), flush=True)
    if sep == '1': ref = A.data.copy()
    else: print('max relative difference of the two paths at C4: %.2e' % (np.abs(ref - A.data).max() / np.abs(A.data).max()))
    del A
PY
