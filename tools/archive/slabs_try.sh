#!/bin/bash
# strong-scaling slabs of the C4 patch, one after the other on this one GPU (--emulate r/W): per-slab median / max step time
cd "$GRAFT_REPO_ROOT"
one() { timeout 300 python bench.py "$@" --no-cpu-baseline --no-api-call --steps 8 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', round(d['step_ms']['median'], 3), round(d['step_ms']['max'], 3), d['roofline']['kernel_ms'], 'cold', d['cold_ms'])
"; }
one
for W in ${1:-8 4 2}; do for ((r=0; r<W; r++)); do one --emulate $r/$W; done; done
