#!/bin/bash
# C4 kernel times for variant builds of the library (tools/buildvar.sh): var_try.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
    lib=$PWD/pyiga_amd/libigx_$v.so
    [ "$v" = base ] && lib=$PWD/pyiga_amd/libigx.so
    echo "== $v"
    IGX_LIB=$lib timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 6 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
done
