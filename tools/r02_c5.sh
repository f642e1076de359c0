#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for c in c5 c2 c1; do
timeout 300 python bench.py --config $c --no-cpu-baseline --steps 5 --warmup 1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$c', d['step_ms']['median'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
done
IGX_PATH=unfused IGX_GEOA=0 timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | cut -c1-100
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "convdiff or final_stage or fixtures or golden or general_form or full_size or fused_equals or fullsize" 2>&1 | tail -3
