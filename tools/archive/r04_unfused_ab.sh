#!/bin/bash
# the stage-kernel chain (IGX_PATH=unfused IGX_GEOA=0) with two builds of the library in one session: tools/r04_unfused_ab.sh libA libB
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in "$@"; do
    lib=$PWD/pyiga_amd/libigx_$v.so; [ "$v" = base ] && lib=$PWD/pyiga_amd/libigx.so
    echo "== $v"
    IGX_LIB=$lib IGX_PATH=unfused IGX_GEOA=0 timeout 300 python bench.py --config c4 --no-cpu-baseline --no-api-call --steps 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
