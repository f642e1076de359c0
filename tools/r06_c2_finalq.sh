#!/bin/bash
# round 6: shapes of k_final_q (waves per block, K lines in flight) for the 2D chain at C2 -- device time of the chain per variant library
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
for v in base qd1 qd2 qw2 qw8; do
    lib=$PWD/pyiga_amd/libigx_$v.so
    [ "$v" = base ] && lib=$PWD/pyiga_amd/libigx.so
    IGX_LIB=$lib timeout 300 python bench.py --config c2 --no-cpu-baseline --no-api-call --steps 50 --warmup 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', 'wall', round(d['ms_per_step'], 4), 'device', d['step_ms'])
    else: print(l.rstrip()[-200:])
"
done
done
