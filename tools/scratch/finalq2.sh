#!/bin/bash
run() { echo "== $*"; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"; }
run IGX_X=0
for n in "$@"; do run IGX_LIB=$PWD/pyiga_amd/csrc/build/libigx_$n.so; run IGX_LIB=$PWD/pyiga_amd/csrc/build/libigx_$n.so IGX_FINALQ_WAVES=4096;  done
