#!/bin/bash
# repeated, interleaved timing of variant libraries (memory-bound kernels vary a lot from run to run): var_rep.sh REPS name...
cd "$GRAFT_REPO_ROOT"
reps=$1; shift
for ((i=0; i<reps; i++)); do
  for v in "$@"; do
    lib=$PWD/pyiga_amd/libigx_$v.so; [ "$v" = base ] && lib=$PWD/pyiga_amd/libigx.so
    IGX_LIB=$lib timeout 300 python bench.py --config c4 --no-cpu-baseline --no-api-call --steps 8 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['ms_per_step'], json.dumps(d['roofline']['kernel_ms']))
"
  done
done | python -c "
import sys, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for l in sys.stdin:
    v, ms, k = l.split(' ', 2)
    acc[v]['chain'].append(float(ms))
    for kk, vv in json.loads(k).items(): acc[v][kk].append(vv)
for v, d in acc.items():
    print(v, {k: (round(min(x), 3), round(sorted(x)[len(x) // 2], 3)) for k, x in d.items()}, '(min, median)')
"
