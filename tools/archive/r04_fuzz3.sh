#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python tools/fuzz_convdiff.py 50 31 2>&1 | grep -E "FAIL|worst"
timeout 1500 python tools/fuzz_paths.py 60 33 - 5 2>&1 | grep -E "FAIL|all .* cases|worst" | tail -3
IGX_GEOA=mfma timeout 900 python tools/fuzz_paths.py 40 35 - 4 2>&1 | grep -E "FAIL|all .* cases|worst" | tail -3
timeout 900 python tools/fuzz_rhs.py 60 37 2>&1 | grep -E "FAIL|worst"
