#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/fuzz_rhs.py 40 4 2>&1 | tail -44
timeout 1200 python tools/fuzz_paths.py 60 9 - 5 2>&1 | tail -5
timeout 900 python tools/fuzz_convdiff.py 30 8 2>&1 | tail -4
IGX_PLACEMENT_TRIES=3 IGX_GEOA=mfma timeout 900 python tools/fuzz_paths.py 30 12 - 4 2>&1 | tail -3
