#!/bin/bash
# k_geoA with parts of its work switched off (timing experiment, wrong results by construction): needs the ablation build
#   make -C pyiga_amd/csrc ablate     ->  pyiga_amd/libigx_ablate.so (the shipped library has no such switches)
# IGX_GEOA_DBG bits: 1 no geometry, 2 no K1 stores, 4 no sweep arithmetic.   tools/geoa_ablate.sh [config ...]  (default c4)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for cfg in ${@:-c4}; do
for d in 0 1 2 4 3 5 6 7; do
    echo "== $cfg IGX_GEOA_DBG=$d"
    IGX_LIB=$PWD/pyiga_amd/libigx_ablate.so IGX_GEOA_DBG=$d timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['roofline']['kernel_ms'])
"
done
done 2>&1 | tee gpurun_out/r06_geoa_ablate.txt
