#!/bin/bash
# quick C4 timing (+ optional env passthrough) and the fused/unfused parity subset
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --config ${1:-c4} --no-cpu-baseline --steps 10 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['step_ms'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_equals or fullsize_vs_reference or row_slabs or tiny or fixtures or full_size or repeat or ablation" 2>&1 | tail -3
