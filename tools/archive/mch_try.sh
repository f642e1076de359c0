#!/bin/bash
# k_bf2 of C4 row slabs against the number of mid-axis chunks (ablation build: IGX_BF_MCHUNKS), one box, one session
cd "$GRAFT_REPO_ROOT"
for e in 1/4 0/2 0/8; do for m in 0 1 2 3 4; do
  if [ $m = 0 ]; then unset IGX_BF_MCHUNKS; else export IGX_BF_MCHUNKS=$m; fi
  IGX_LIB=$PWD/pyiga_amd/libigx_ablate.so python bench.py --emulate $e --no-cpu-baseline --steps 6 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$e m=$m', round(d['step_ms']['median'],3), d['roofline']['kernel_ms'])
"; done; done
