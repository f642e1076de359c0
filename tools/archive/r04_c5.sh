#!/bin/bash
# round 4: the convection-diffusion chain with the geometry inside the axis-0 sweep (k_geoA, non-symmetric) against the
# field kernel + k_stageA (IGX_GEOA=0), same box, + the convdiff parity tests
cd "$GRAFT_REPO_ROOT"
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['step_ms'], d['roofline']['kernel_ms'], d['config']['path'])
    else: print(l.rstrip()[-300:])
"; }
echo "== k_geoA (non-symmetric)"; timeout 300 python bench.py --config c5 --no-cpu-baseline --no-api-call --steps 6 2>&1 | line
echo "== field kernel + k_stageA"; IGX_GEOA=0 timeout 300 python bench.py --config c5 --no-cpu-baseline --no-api-call --steps 6 2>&1 | line
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "convdiff or c5" 2>&1 | tail -8
