#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in base ngw8 ngw2 base; do
    lib=$PWD/pyiga_amd/libigx_$v.so; [ "$v" = base ] && lib=$PWD/pyiga_amd/libigx.so
    echo "== $v"
    IGX_LIB=$lib timeout 300 python bench.py --config c5 --no-cpu-baseline --no-api-call --steps 6 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'])
    else: print(l.rstrip()[-300:])
"
done
timeout 300 python bench.py --op rhs --steps 10 2>&1 | tail -1 | cut -c1-300
