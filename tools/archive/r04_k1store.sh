#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/var_try.sh base nstw3 base nstw3
python bench.py --config c5 --no-cpu-baseline --no-api-call --steps 6 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'])
"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "geometry_in_sweep or row_slabs or fixtures or golden_matrices or convdiff" 2>&1 | tail -3
