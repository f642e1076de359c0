#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for pad in 0 32 64 160 576 4128; do
  for d in 0 5; do
    echo "== c4 IGX_K1PAD=$pad DBG=$d"
    IGX_K1PAD=$pad IGX_GEOA_DBG=$d timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
  done
done
IGX_K1PAD=64 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_equals or fullsize_vs_reference_3d or row_slabs" 2>&1 | tail -3
