#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for d in 0 1 2 4 3 5 6 7; do
    echo "== c4 IGX_GEOA_DBG=$d"
    IGX_GEOA_DBG=$d timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
done
