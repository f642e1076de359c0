#!/bin/bash
cd "$GRAFT_REPO_ROOT"
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['ms_per_step'], 3), d['roofline']['kernel_ms'], d.get('placement'), 'cold_ms', d['cold_ms'])
    else: print(l.rstrip()[-300:])
"; }
for rep in 1 2 3; do
echo "== tries 4"; timeout 300 python bench.py --no-cpu-baseline --no-api-call --steps 10 2>&1 | line
echo "== tries 1"; timeout 300 python bench.py --no-cpu-baseline --no-api-call --steps 10 --placement-tries 1 2>&1 | line
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "placement or full_size_c4" 2>&1 | tail -4
