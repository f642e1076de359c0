#!/bin/bash
# round 4: A/B of the matrix-core axis-0 sweep in k_geoA against the vector form (IGX_GEOA=valu), same box, + parity subset
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_ub
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['step_ms'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"; }
for rep in 1 2; do
echo "== matrix cores"; timeout 300 python bench.py --config c4 --no-cpu-baseline --no-api-call --steps 10 2>&1 | line
echo "== vector form"; IGX_GEOA=valu timeout 300 python bench.py --config c4 --no-cpu-baseline --no-api-call --steps 10 2>&1 | line
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_equals or fullsize_vs_reference or row_slabs or tiny or fixtures or full_size or repeat or ablation or golden_matrices" 2>&1 | tail -5
