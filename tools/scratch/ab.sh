#!/bin/bash
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],2), d['roofline']['kernel_ms'])"; }
for i in 1 2 3; do run IGX_FINAL=q; run IGX_FINAL=valu; done
