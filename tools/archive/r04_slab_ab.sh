#!/bin/bash
# slabs of the strong-scaling split of C4 with two builds of the library in one session: tools/r04_slab_ab.sh libA libB
cd "$GRAFT_REPO_ROOT"
for sl in 3/8 0/8 1/4; do for v in "$@" "$@"; do
    lib=$PWD/pyiga_amd/libigx_$v.so; [ "$v" = base ] && lib=$PWD/pyiga_amd/libigx.so
    echo -n "slab $sl $v: "
    IGX_LIB=$lib timeout 300 python bench.py --emulate $sl --no-cpu-baseline --steps 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['roofline']['kernel_ms'])"
done; done
