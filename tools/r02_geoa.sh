#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fullsize or fused_equals or row_slabs or reference_fixtures or golden_matrices or properties_larger or tiny or full_size" > gpurun_out/pytest_geoa.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_geoa.log
for c in c4 c3 c5; do
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
