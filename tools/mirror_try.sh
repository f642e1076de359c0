#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in -1 0 1 2; do
  echo "== IGX_MIRROR=$v"
  IGX_MIRROR=$v timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 8 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
done
