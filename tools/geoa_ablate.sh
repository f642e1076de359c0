#!/bin/bash
# k_geoA with parts of its work switched off (timing experiment, wrong results by construction): needs the ablation build
#   make -C pyiga_amd/csrc ablate     ->  pyiga_amd/libigx_ablate.so (the shipped library has no such switches)
cd "$GRAFT_REPO_ROOT"
for d in 0 1 2 4 3 5 6 7; do
    echo "== c4 IGX_GEOA_DBG=$d"
    IGX_LIB=$PWD/pyiga_amd/libigx_ablate.so IGX_GEOA_DBG=$d timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['roofline']['kernel_ms'])
"
done
