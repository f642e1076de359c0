#!/bin/bash
# timing experiments on the fused stage (C4): ablation masks of k_bf
for d in ${@:-0 1 2 3 4 8 12 7 15}; do
  echo -n "IGX_BF_DBG=$d  "
  IGX_PATH=fused IGX_BF_DBG=$d python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['kernel_ms'])"
done
