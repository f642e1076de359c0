#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for c in c4 c3; do
  for g in 1 0; do
    echo "== $c IGX_GEOA=$g"
    IGX_GEOA=$g timeout 300 python bench.py --config $c --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['step_ms'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
  done
done
