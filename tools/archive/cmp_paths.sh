#!/bin/bash
# fused vs unfused chain per BASELINE config (device time, ms)
for c in ${@:-c1 c2 c3 c4}; do
  for pth in fused unfused; do
    echo -n "$c $pth  "
    IGX_PATH=$pth python bench.py --config $c --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['roofline']['chain_ms'],4), r['roofline']['kernel_ms'])"
  done
done
