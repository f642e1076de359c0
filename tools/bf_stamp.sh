#!/bin/bash
cd "$GRAFT_REPO_ROOT"
IGX_LIB=$PWD/pyiga_amd/libigx_stamp.so timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 1 --warmup 1 2>&1 | grep "stamp" | tail -40
