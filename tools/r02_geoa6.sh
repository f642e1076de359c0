#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for d in 0 2 8; do
    echo "== c4 stamp DBG=$d"
    IGX_LIB=$PWD/pyiga_amd/libigx_stamp.so IGX_GEOA_DBG=$d timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 1 --warmup 1 2>&1 | grep "stamp" | tail -8
done
