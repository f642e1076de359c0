#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for l in m44 m33; do
  echo "== $l"
  IGX_LIB=$PWD/pyiga_amd/libigx_$l.so timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
done
