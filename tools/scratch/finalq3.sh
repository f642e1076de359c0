#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for a in "$@"; do run $a; done
